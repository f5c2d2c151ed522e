"""CPU-only tests: the C-ABI library loads and exports every symbol the header
declares, weight packing, the tracker's host logic (delta bookkeeping, memory
ring, cache protocol) and the delta-sharded multi-rank path over gloo.

The tracker's compute backend is the HIP library in the product; here (no GPU)
the host logic is exercised with a test double built on the CPU oracle."""
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import golden_inputs as gi
from oracle import mft_oracle as O

REPO = Path(__file__).resolve().parents[1]


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


# ---------------------------------------------------------------------------
# boundary
# ---------------------------------------------------------------------------

def test_library_exports_every_header_symbol():
    if not (REPO / "mft_amd" / "libmftx.so").exists():
        subprocess.run(["make", "-C", str(REPO / "mft_amd" / "csrc"), "-j8"], check=True, capture_output=True)
    from mft_amd import _lib
    lib = _lib.load()
    header = (REPO / "include" / "mftx.h").read_text()
    declared = set(re.findall(r"\b(mftx_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mftx_version() == 400
    assert lib.mftx_raft_workspace_bytes(7, 64, 64) > 7 * 4096 * 4096 * 4
    assert lib.mftx_raft_workspace_bytes(0, 64, 64) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mft_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setenv("MFTX_LIB", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MftxError):
        _lib.load()
    monkeypatch.delenv("MFTX_LIB")
    monkeypatch.setattr(_lib, "_lib", None)
    _lib.load()


def test_ops_reject_cpu_tensors():
    from mft_amd import ops
    from mft_amd._lib import MftxError
    z = (torch.zeros(2, 8, 8), torch.zeros(1, 8, 8), torch.zeros(1, 8, 8))
    with pytest.raises(MftxError):
        ops.chain(z, z)


def test_flow_plugin_refuses_cpu():
    from mft_amd.config import Config
    from mft_amd.raft import RAFTWrapper
    with pytest.raises(RuntimeError):
        RAFTWrapper(Config(), device="cpu")


def test_pack_conv_weight_layout():
    from mft_amd.ops import pack_conv_weight
    w = torch.arange(3 * 5 * 1 * 5, dtype=torch.float32).reshape(3, 5, 1, 5)
    p = pack_conv_weight(w)
    assert p.shape == (128, 5, 32)
    assert p[2, 3, 4] == w[2, 4, 0, 3]          # [n][tap][cin]
    assert float(p[3:].abs().sum()) == 0 and float(p[:, :, 5:].abs().sum()) == 0


def test_weight_schema_matches_reference_counts(weights_np):
    from mft_amd.weights import schema
    assert len(schema()) == 187
    n = sum(int(np.prod(s)) for _, s, k in schema())
    assert n == 6905492
    assert weights_np["update_block.gru.convz1.weight"].shape == (128, 384, 1, 5)


def test_pad_amounts_match_oracle():
    from mft_amd.raft import pad_amounts
    for H in range(120, 137):
        for W in (187, 192, 193):
            assert pad_amounts(H, W) == O.pad_amounts(H, W)
    assert pad_amounts(1080, 1920) == (0, 0, 0, 0) and pad_amounts(436, 1024) == (0, 0, 2, 2)


def test_config_semantics(tmp_path):
    from mft_amd.config import Config, load_config
    c = Config()
    assert not c.timers_enabled and not c.a.b.c
    c.x = 3
    d = Config(); d.x = 4; d.y = 5
    c.merge(d)
    assert c.x == 4 and c.y == 5
    cfg = load_config(REPO / "configs" / "MFT_cfg.py")
    assert cfg.deltas[0] == np.inf and cfg.deltas[1:] == [1, 2, 4, 8, 16, 32]
    assert cfg.occlusion_threshold == 0.02 and cfg.flow_config.flow_iters == 12


# ---------------------------------------------------------------------------
# tracker host logic with an oracle-backed test double
# ---------------------------------------------------------------------------

class OracleBackend:
    @staticmethod
    def chain(L, R):
        return O.chain(L, R)

    @staticmethod
    def select(cands, thr):
        f, o, s, idx = O.select(cands, thr)
        return f, o, s, idx.to(torch.int8)

    @staticmethod
    def chain_select(Ls, Rs, thr):
        from mft_amd.MFT import is_packed, unpack_planes
        Rs = [unpack_planes(r) if is_packed(r) else r for r in Rs]          # the multi-rank path hands packed results
        return OracleBackend.select([O.chain(l, r) for l, r in zip(Ls, Rs)], thr)


class StubFlower:
    def __init__(self):
        self.calls = []

    def compute_flow(self, src_img, dst_img, mode="flow", init_flow=None, **kw):
        l, r = gi.decode_id(src_img), gi.decode_id(dst_img)
        self.calls.append((l, r))
        flow, occl, sigma = gi.stub_flowou(l, r)
        return T(flow), {"occlusion": T(occl), "sigma": T(sigma), "debug": None}


def make_tracker(flower, deltas=(np.inf, 1, 2, 4, 8, 16, 32), **extra):
    from mft_amd.config import Config
    from mft_amd.MFT import MFT
    c = Config()
    c.deltas = list(deltas)
    c.occlusion_threshold = 0.02
    c.flow_config = Config()
    c.flow_config.of_class = lambda cfg: flower
    for k, v in extra.items():
        setattr(c, k, v)
    return MFT(c, backend=OracleBackend(), device="cpu")


@pytest.mark.parametrize("tag,start,direction", [("fwd", 0, 1), ("bwd", gi.SEQ_FRAMES - 1, -1)])
def test_tracker_bookkeeping_vs_reference(golden_dir, tag, start, direction):
    g = np.load(golden_dir / "sequence_stub.npz")
    fl = StubFlower()
    tr = make_tracker(fl)
    tr.init(gi.id_image(start), start_frame_i=start, time_direction=direction)
    for step in range(1, gi.SEQ_FRAMES):
        fid = start + direction * step
        fl.calls.clear()
        res = tr.track(gi.id_image(fid)).result
        keys = [k for k in g[f"{tag}_memory_keys"][step - 1].tolist() if k >= 0]
        assert sorted(tr.memory.keys()) == keys
        # requested pairs: every delta that does not reach before the start, inf -> start, deduplicated
        want = []
        for d in [np.inf, 1, 2, 4, 8, 16, 32]:
            left = fid - d * direction
            before = left < start if direction > 0 else left > start
            if before:
                if not np.isinf(d):
                    continue
                left = start
            if int(left) not in want:
                want.append(int(left))
        assert sorted(l for l, r in fl.calls) == sorted(want) and all(r == fid for l, r in fl.calls)
        cs = np.concatenate([gi.checksum(res.flow.numpy()), gi.checksum(res.occlusion.numpy()),
                             gi.checksum(res.sigma.numpy())])
        assert np.allclose(cs, g[f"{tag}_checksums"][step - 1], rtol=2e-3, atol=1.0), step


def test_flow_cache_protocol():
    """finite deltas go through cache.read/.write, delta=inf does not (unless
    cache_delta_infinity); a failing cache means recompute (MFT/MFT.py:99-102,214-219)."""
    class Cache:
        def __init__(self):
            self.store, self.reads, self.writes = {}, [], []

        def read(self, l, r):
            self.reads.append((l, r))
            if (l, r) == (2, 3):
                raise IOError("broken entry")
            return self.store.get((l, r), (None, None, None))

        def write(self, l, r, f, o, s):
            self.writes.append((l, r))
            self.store[(l, r)] = (f, o, s)

    fl, cache = StubFlower(), Cache()
    tr = make_tracker(fl, deltas=(np.inf, 1, 2))
    tr.init(gi.id_image(0), flow_cache=cache)
    for i in range(1, 5):
        tr.track(gi.id_image(i))
    assert (0, 4) not in cache.reads and (0, 3) not in cache.writes      # delta=inf bypasses the cache
    assert (3, 4) in cache.writes and (2, 4) in cache.writes
    assert (2, 3) in cache.reads and (2, 3) in fl.calls                  # broken read -> recomputed
    # second run over the same frames: everything finite is served from the cache
    fl2 = StubFlower()
    tr2 = make_tracker(fl2, deltas=(np.inf, 1, 2))
    tr2.init(gi.id_image(0), flow_cache=cache)
    for i in range(1, 5):
        tr2.track(gi.id_image(i))
    assert all(l == 0 or (l, r) == (2, 3) for l, r in fl2.calls), fl2.calls
    # with cache_delta_infinity the inf pair is cached too
    fl3, cache3 = StubFlower(), Cache()
    tr3 = make_tracker(fl3, deltas=(np.inf, 1), cache_delta_infinity=True)
    tr3.init(gi.id_image(0), flow_cache=cache3)
    tr3.track(gi.id_image(1)); tr3.track(gi.id_image(2))
    assert (0, 2) in cache3.writes


def test_get_flowou_with_cache_and_chain_results_api():
    from mft_amd.MFT import get_flowou_with_cache
    fl = StubFlower()
    r = get_flowou_with_cache(fl, gi.id_image(3), gi.id_image(5))
    assert r.flow.shape == (2, gi.SEQ_H, gi.SEQ_W) and fl.calls == [(3, 5)]
    with pytest.raises(AssertionError):
        get_flowou_with_cache(fl, gi.id_image(3), gi.id_image(5), read_cache=True)   # ids required


def test_keep_result_on_device_flag():
    tr = make_tracker(StubFlower(), deltas=(np.inf, 1), keep_result_on_device=True)
    tr.init(gi.id_image(0))
    m = tr.track(gi.id_image(1))
    stored = tr.memory[1]["result"]
    assert m.result is not stored and torch.equal(m.result.flow, stored.flow)
    m.result.flow.add_(1.0)                      # a consumer that edits its result in place ...
    assert not torch.equal(m.result.flow, stored.flow)   # ... does not touch tracker state


def test_missing_checkpoint_is_an_error():
    """A configured but absent checkpoint raises like the reference's torch.load (MFT/raft.py:20-21);
    synthetic weights need the explicit opt-in model=None + integer seed."""
    from mft_amd.config import Config
    from mft_amd.raft import RAFTWrapper
    c = Config()
    c.model = "checkpoints/does-not-exist.pth"
    c.synthetic_weights_seed = 0
    with pytest.raises(FileNotFoundError):
        RAFTWrapper._load_weights(c)
    c.model = None
    c.synthetic_weights_seed = Config()           # "unset"
    with pytest.raises(ValueError):
        RAFTWrapper._load_weights(c)
    c.synthetic_weights_seed = 3
    assert len(RAFTWrapper._load_weights(c)) == 187


def test_track_window_equals_track():
    """track_window on one rank is track() frame by frame."""
    a, b = make_tracker(StubFlower()), make_tracker(StubFlower())
    a.init(gi.id_image(0)); b.init(gi.id_image(0))
    ra = [a.track(gi.id_image(i)).result for i in range(1, 7)]
    rb = [m.result for m in b.track_window([gi.id_image(i) for i in range(1, 4)])]
    rb += [m.result for m in b.track_window([gi.id_image(i) for i in range(4, 7)])]
    for x, y in zip(ra, rb):
        assert torch.equal(x.flow, y.flow) and torch.equal(x.sigma, y.sigma)
    assert sorted(a.memory) == sorted(b.memory) and a.current_frame_i == b.current_frame_i == 6


# ---------------------------------------------------------------------------
# multi-rank (gloo, world_size 2): sharded == single-rank, bitwise
# ---------------------------------------------------------------------------

def test_shard_plan():
    from mft_amd.dist import frame_owner, shard_indices, split_units
    assert split_units(7, 8) == [(i, 1) for i in range(7)] + [(7, 0)]
    assert split_units(7, 2) == [(0, 4), (4, 3)] and split_units(112, 8) == [(14 * r, 14) for r in range(8)]
    assert shard_indices(7, 8, 7) == [] and shard_indices(7, 2, 1) == [4, 5, 6]
    for K in (1, 7, 20, 140):
        for G in (1, 2, 4, 8):
            sh = split_units(K, G)
            assert sorted(sum((list(range(o, o + c)) for o, c in sh), [])) == list(range(K))
            assert max(c for _, c in sh) - min(c for _, c in sh) <= 1
    assert [frame_owner(j, 4) for j in range(6)] == [0, 1, 2, 3, 0, 1]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_window_sharding_gloo(tmp_path, world):
    """World size 2, 4 and 8 over gloo: the per-frame mode (L = 1; at world 4 some ranks own no unit of a
    frame), look-ahead windows (L = 8, ragged last window), the feature-exchanging path (L = 5, also with the
    next window's exchange started early) and the pipelined mode (results one window late: the all-gather
    and the selections of a window overlap the next window's batches) all reproduce the single-rank
    tracker bit for bit, on every rank; with exchanged features no rank encodes a frame another rank owns."""
    script = REPO / "tests" / "dist_worker.py"
    port = str(29609 + world)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = tmp_path / "out"
    out.mkdir()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", port, str(script), str(out)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    single = np.load(out / "single.npz")
    for mode in ("L1", "L8", "L1x", "L5x", "L5p", "L5d"):
        rk = [np.load(out / f"rank{r}_{mode}.npz") for r in range(world)]
        for k in single.files:
            for r in range(world):                                      # replicas stay identical, and equal to the unsharded run
                assert np.array_equal(rk[r][k], single[k]), (mode, r, k)
    for mode in ("L1x", "L5x", "L5p", "L5d"):
        st = [np.load(out / f"rank{r}_{mode}.npz") for r in range(world)]
        # every window frame was encoded exactly once across the ranks (also when the next window's exchange is
        # started early, L5p), none outside the exchange: a window with at least half as many frames as ranks has whole frames
        # encoded by their owners, a shorter one -- the per-frame mode L1x, a ragged last window -- one NETWORK of a frame per rank
        # (two halves = one frame)
        frames = int(st[0]["_frames"])
        assert sum(float(s["_encoded"]) for s in st) == frames, mode
        assert all(int(s["_local"]) == 0 for s in st), mode
        units = [int(s["_my_units"]) for s in st]
        assert max(units) - min(units) <= int(st[0]["_windows"])
    # the flow cache in the sharded path (owner-resolved): a cold run equals the uncached tracker, a warm run is served from
    # the owners' caches (no finite-delta pair recomputed) and equals the single-rank tracker's warm run -- which chains the
    # quantised entries -- bit for bit, on every rank
    single_warm = np.load(out / "single_warm.npz")
    ck = [np.load(out / f"rank{r}_cache.npz") for r in range(world)]
    for r in range(world):
        for k in single.files:
            assert np.array_equal(ck[r]["cold_" + k], single[k]), ("cold", r, k)
            assert np.array_equal(ck[r]["warm_" + k], single_warm[k]), ("warm", r, k)
        assert int(ck[r]["_recomputed"]) == 0, r
    assert sum(int(c["_hits"]) for c in ck) == sum(int(c["_writes"]) for c in ck) > 0
    assert not all(np.array_equal(single_warm[k], single[k]) for k in single.files)      # (the quantisation shows)


def oracle_flow_cache():
    """FlowCache whose disk tier quantises with the ORACLE instead of the HIP kernels (no GPU here);
    the container code (pickle + PNG, C unfilter) is the product's."""
    from mft_amd.io import FlowCache
    from mft_amd import flowou_codec as fc

    class OracleCodecCache(FlowCache):
        def _save(self, path, val):
            planes = [val[0][0], val[0][1], val[1][0], val[2][0]]
            fc.pack_flowou_X16(path, [O.quantize_u16(p.numpy()) for p in planes])

        def _load(self, path):
            fx, fy, oc, sg = (torch.from_numpy(O.dequantize_u16(q, lo, hi)) for q, lo, hi in fc.unpack_flowou_X16(path))
            return torch.stack([fx, fy]), oc[None], sg[None]

    return OracleCodecCache


def test_flow_cache_tiers(tmp_path):
    FlowCache = oracle_flow_cache()
    f = lambda v: (torch.full((2, 4, 4), float(v)), torch.zeros(1, 4, 4), torch.ones(1, 4, 4))  # noqa: E731
    one = 4 * 4 * 4 * 4                      # bytes of one (flow, occl, sigma) triple
    c = FlowCache(tmp_path / "c", max_RAM_MB=2 * one / 1e6, max_GPU_RAM_MB=one / 1e6, device="cpu")
    assert c.read(0, 1) == (None, None, None)
    for i in range(5):
        c.write(i, i + 1, *f(i))
    assert len(c.gpu_ram_cache) == 1 and len(c.ram_cache) == 2 and len(list((tmp_path / "c").glob("*.flowouX16.pkl"))) == 2
    for i in range(5):
        assert float(c.read(i, i + 1)[0][0, 0, 0]) == i
    assert c.backup_to_disk() == 3
    d = FlowCache(tmp_path / "c", device="cpu")
    assert d.load_from_disk() == 5 and float(d.read(3, 4)[0][0, 0, 0]) == 3
    c.clear()
    assert c.read(0, 1) == (None, None, None)
    # drives the tracker like the reference's FlowCache does
    fl = StubFlower()
    cache = FlowCache(None, device="cpu")
    tr = make_tracker(fl, deltas=(np.inf, 1, 2))
    tr.init(gi.id_image(0), flow_cache=cache)
    for i in range(1, 4):
        tr.track(gi.id_image(i))
    assert (2, 3) in cache.gpu_ram_cache and (0, 3) not in cache.gpu_ram_cache


def test_point_tracking_adapter():
    from mft_amd.point_tracking import convert_to_point_tracking
    from mft_amd.results import FlowOUTrackingResult
    flow = torch.zeros(2, 16, 24); flow[0] = 2.0; flow[1] = -1.0
    occl = torch.zeros(1, 16, 24); occl[0, 8:, :] = 1.0
    r = FlowOUTrackingResult(flow, occl, torch.zeros(1, 16, 24))
    q = np.array([[3.0, 4.0], [10.0, 12.0]], np.float32)
    coords, o = convert_to_point_tracking(r, q)
    assert np.allclose(coords, q + [2.0, -1.0]) and np.allclose(o, [0.0, 1.0])


def test_every_raw_s_barrier_is_preceded_by_an_lds_wait():
    """gfx950 has back-off barriers: the compiler inserts no s_waitcnt in front of s_barrier and __builtin_amdgcn_s_barrier() is no
    fence, so a hand-written barrier must wait for its wave's LDS operations itself (round 5: the race behind
    tests/test_gpu_contention.py).  Source-level guard: every raw s_barrier in csrc/ has `s_waitcnt lgkmcnt(0)` just in front."""
    import re
    bad = []
    for path in sorted((REPO / "mft_amd" / "csrc").glob("*.h*")):
        lines = path.read_text().splitlines()
        for i, line in enumerate(lines):
            if "__builtin_amdgcn_s_barrier()" in line and not line.lstrip().startswith("//"):
                before = "\n".join(lines[max(0, i - 3): i])
                if not re.search(r's_waitcnt lgkmcnt\(0\)', before):
                    bad.append(f"{path.name}:{i + 1}")
    assert not bad, bad


def test_chain_kernels_are_built_without_packed_fp32_code():
    """Round 5, the chain race: the SLP vectoriser's packed-fp32 code in the chain + selection kernels lost lanes 48..63 of one result
    register whenever another process shared the GPU (profiles/r5q_chain_race.txt).  Build-level guard: with the flags of the
    Makefile's chain.o rule -- and of tools/build_tuning.sh -- the device code of chain.hip contains no v_pk_* instruction."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("no hipcc")
    mk = (REPO / "mft_amd" / "csrc" / "Makefile").read_text()
    m = re.search(r"^chain\.o:.*\n\t\$\(HIPCC\) \$\(CXXFLAGS\) (.*?) -c \$< -o \$@", mk, re.M)
    assert m, "chain.o rule not found"
    nopk = re.search(r"^NOPK := (.*)$", mk, re.M).group(1).split()
    extra = [f for tok in m.group(1).split() for f in (nopk if tok == "$(NOPK)" else [tok])]
    assert "-fno-slp-vectorize" in extra and "-ffp-contract=off" in extra, extra
    assert '"chain.hip|chain.o|-ffp-contract=off -fno-slp-vectorize $NOPK"' in (REPO / "tools" / "build_tuning.sh").read_text()
    cxx = re.search(r"^CXXFLAGS := (.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    out = subprocess.run([hipcc, *cxx, *extra, "-I", str(REPO / "include"), "-S", "--cuda-device-only", "-o", "-",
                          str(REPO / "mft_amd" / "csrc" / "chain.hip")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "chain_select_packed_kernel" in out.stdout
    assert not re.search(r"\bv_pk_\w+", out.stdout), re.findall(r"\bv_pk_\w+", out.stdout)[:5]


@pytest.mark.parametrize("unit", ["corr", "corr_ondemand", "upsample", "encoder", "raft_engine", "codec"])
def test_masked_gather_units_carry_no_packed_fp32_code(unit):
    """Round 6: the chain race needed packed-fp32 arithmetic AND EXEC-masked gathers in one instruction
    stream (profiles/r5q_chain_race.txt) and its exact hazard is not pinned down -- so every translation unit that gathers behind
    bounds tests is built with the packed-fp32 target feature off.  With the flags of the Makefile's rule for the unit the device
    code has s_and_saveexec regions (the masked gathers are there) and not one v_pk_* instruction."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("no hipcc")
    mk = (REPO / "mft_amd" / "csrc" / "Makefile").read_text()
    nopk = re.search(r"^NOPK := (.*)$", mk, re.M).group(1).split()
    assert nopk == ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
    if unit == "codec":
        rule = re.search(r"^codec\.o:.*\n\t\$\(HIPCC\) \$\(CXXFLAGS\) (.*?) -c \$< -o \$@", mk, re.M)
    else:
        rule = re.search(r"^corr\.o corr_ondemand\.o upsample\.o encoder\.o raft_engine\.o:.*\n\t\$\(HIPCC\) \$\(CXXFLAGS\) (.*?) -c \$< -o \$@", mk, re.M)
    assert rule and "$(NOPK)" in rule.group(1), "Makefile rule without NOPK"
    extra = [f for tok in rule.group(1).split() for f in (nopk if tok == "$(NOPK)" else [tok])]
    tune = (REPO / "tools" / "build_tuning.sh").read_text()
    assert ('"codec.hip|codec.o|-ffp-contract=off $NOPK"' in tune if unit == "codec" else
            re.search(r"for f in [a-z_ ]*\b%s\b[a-z_ ]*; do jobs\+=\(\"\$f\.hip\|\$f\.o\|\$NOPK\"\)" % unit, tune)), "tuning build without NOPK"
    cxx = re.search(r"^CXXFLAGS := (.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    out = subprocess.run([hipcc, *cxx, *extra, "-I", str(REPO / "include"), "-S", "--cuda-device-only", "-o", "-",
                          str(REPO / "mft_amd" / "csrc" / f"{unit}.hip")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "s_and_saveexec" in out.stdout
    assert not re.search(r"\bv_pk_\w+", out.stdout), re.findall(r"\bv_pk_\w+", out.stdout)[:5]


def test_import_after_hip_initialisation_warns(monkeypatch):
    """GPU_MAX_HW_QUEUES is read by the HIP runtime when it initialises: set at package import it is a silent no-op if a GPU call came
    first (hygiene).  The package now says so -- and records what became of the setting in mft_amd.HW_QUEUES."""
    import importlib
    import warnings
    import torch
    import mft_amd
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        importlib.reload(mft_amd)
    assert any("GPU_MAX_HW_QUEUES" in str(x.message) for x in w)
    assert mft_amd.HW_QUEUES == {"explicit": False, "too_late": True, "value": "8"}
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "6")                 # an explicit setting wins and is not second-guessed
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        importlib.reload(mft_amd)
    assert not w and mft_amd.HW_QUEUES == {"explicit": True, "too_late": False, "value": "6"}
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: False)
    importlib.reload(mft_amd)
    assert mft_amd.HW_QUEUES == {"explicit": False, "too_late": False, "value": "8"}


def test_variant_library_exports_the_same_symbols():
    """mft_amd/libmftx_lfwide.so (the fused lookup with 16-byte gathers, `make lfwide`) is a drop-in for libmftx.so: every symbol of
    include/mftx.h resolves in it too."""
    import ctypes
    lib = REPO / "mft_amd" / "libmftx_lfwide.so"
    if not lib.exists():
        pytest.skip("variant library not built (make -C mft_amd/csrc lfwide)")
    names = re.findall(r"\b(mftx_\w+)\s*\(", (REPO / "include" / "mftx.h").read_text())
    h = ctypes.CDLL(str(lib))
    missing = [n for n in sorted(set(names)) if not hasattr(h, n)]
    assert not missing, missing


def test_package_level_names_resolve():
    """`from mft_amd import MFT` gives the tracker CLASS (round 6: the lazy attribute hook recursed for it), and the submodule path of the
    reference-style import keeps working."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c", "from mft_amd import MFT, chain_results, get_flowou_with_cache, RAFTWrapper, FlowOUResult\n"
                          "from mft_amd.MFT import MFT as M2\nimport inspect\nassert inspect.isclass(MFT) and M2 is MFT and inspect.isclass(RAFTWrapper)\nprint('ok')"],
                         capture_output=True, text=True, cwd=str(REPO), timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-1500:]
