// Native runtime for the RAFT refinement loop: owns the packed-weight table,
// carves the caller's workspace, and enqueues the whole iteration sequence
// (core/raft.py:141-226) on one HIP stream -- ~14 launches per iteration, no
// host round trips, batched over P image pairs (M = P*h*w cells).
#include "common.h"
#include "profile.h"
#include "motion_front.h"
#include "graph_cache.h"
#include <cstdlib>
#include <new>

namespace mftx {

// ---------------------------------------------------------------------------
// small glue kernels
// ---------------------------------------------------------------------------

// hx[m] = [net | inp | (motion: filled later)], coords1 = pixel grid
// (core/raft.py:146-151; coords_grid core/utils/utils.py:115-118)
// hf (split arithmetic with split-form activations): hx is written in split form (common.h) and h additionally as
// fp32 into hf [M][128], the copy the GRU's gate algebra reads
// gathered != 0: pair b's net / inp maps at netp.p[b] / inpp.p[b] (mftx_raft_refine_gather) instead of net + b N 128
struct InitGather { int on; PairPtrs netp, inpp; };
__global__ void init_state_kernel(const float *__restrict__ net, const float *__restrict__ inp,
                                  const float *__restrict__ flow_init, float *hx, float *hf, float *coords1, int M, int h,
                                  int w, InitGather ga) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over M*64 float4 slots
    if (i >= (long long)M * 64) return;
    const int m = (int)(i >> 6), q = (int)(i & 63);
    if (ga.on) {                                     // (this pair's maps: the index below becomes the cell inside the pair)
        const int b = m / (h * w);
        net = ga.netp.p[b] - (long long)b * h * w * 128;
        inp = ga.inpp.p[b] - (long long)b * h * w * 128;
    }
    const float4 v = (q < 32) ? reinterpret_cast<const float4 *>(net)[(long long)m * 32 + q]
                              : reinterpret_cast<const float4 *>(inp)[(long long)m * 32 + (q - 32)];
    if (hf != nullptr) {
        store_split4v(hx + (long long)m * 384, 4 * q, v);
        if (q < 32) reinterpret_cast<float4 *>(hf + (long long)m * 128)[q] = v;
    } else {
        reinterpret_cast<float4 *>(hx + (long long)m * 384)[q] = v;
    }
    if (q == 0) {
        const int rem = m % (h * w);
        float cx = (float)(rem % w), cy = (float)(rem / w);
        if (flow_init != nullptr) {                 // coords1 = coords0 + flow_init (core/raft.py:153-154)
            cx += flow_init[2 * (long long)m];
            cy += flow_init[2 * (long long)m + 1];
        }
        coords1[2 * (long long)m] = cx;
        coords1[2 * (long long)m + 1] = cy;
    }
}

// two strips per 256-thread block
__global__ __launch_bounds__(256) void convf1_kernel(ConvF1Args q) {
    __shared__ __attribute__((aligned(16))) float patch[2][7][48];
    const int half = threadIdx.x >> 7;
    convf1_strip(q, blockIdx.x * 2 + half, patch[half], threadIdx.x & 127);
}

// Lookup and convf1 read nothing but coords1 and the pyramid / the weights, and they stress different
// parts of the chip (HBM gathers vs VALU): one launch carries both, block types interleaved in the ratio
// of their counts so that every CU holds both kinds at once.  Used by the refinement loop; the per-kernel
// timing pass and the per-op export launch them separately.
__global__ __launch_bounds__(256) void lookup_convf1_kernel(LookupArgs lp, ConvF1Args q, int lookup_blocks, int f1_blocks) {
    __shared__ __attribute__((aligned(16))) float patch[2][7][48];
    const int total = lookup_blocks + f1_blocks;
    const int b = blockIdx.x;
    const int f_lo = (int)((long long)b * f1_blocks / total), f_hi = (int)((long long)(b + 1) * f1_blocks / total);
    if (f_hi > f_lo) {
        const int half = threadIdx.x >> 7;
        convf1_strip(q, f_lo * 2 + half, patch[half], threadIdx.x & 127);
    } else {
        lookup_block_body<2>(lp, b - f_lo, lookup_blocks);
    }
}

// OU input [net128 | inp128 | corr324 | flow2 | delta2 | motion128] = 712
// (core/update.py:197), flow = coords1 - grid AFTER the last update
// (core/raft.py:199-206); also emits flow_lr for the upsampler.
// split != 0: hx is in split form and so is ouin -- four channels are one half group (common.h: load_split4 /
// store_split4v; decode + encode reproduces the halves exactly); corr is fp32 in either mode
__global__ void ou_gather_kernel(const float *__restrict__ hx, const float *__restrict__ corr, int ld_corr,
                                 const float *__restrict__ coords1, const float *__restrict__ delta,
                                 float *__restrict__ ouin, float *__restrict__ flow_lr, int M, int h, int w, int split) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over M*178 float4 slots
    if (i >= (long long)M * 178) return;
    const int m = (int)(i / 178), q = (int)(i - (long long)m * 178);
    float4 v;
    if (split) {
        if (q < 64) v = load_split4(hx + (long long)m * 384, 4 * q);
        else if (q < 145) v = reinterpret_cast<const float4 *>(corr + (long long)m * ld_corr)[q - 64];
        else if (q == 145) {
            const int rem = m % (h * w);
            const float fx = coords1[2 * (long long)m] - (float)(rem % w);
            const float fy = coords1[2 * (long long)m + 1] - (float)(rem / w);
            v = make_float4(fx, fy, delta[2 * (long long)m], delta[2 * (long long)m + 1]);
            flow_lr[2 * (long long)m] = fx;
            flow_lr[2 * (long long)m + 1] = fy;
        } else v = load_split4(hx + (long long)m * 384 + 256, 4 * (q - 146));
        store_split4v(ouin + (long long)m * 712, 4 * q, v);
        return;
    }
    if (q < 64) v = reinterpret_cast<const float4 *>(hx + (long long)m * 384)[q];                 // net, inp
    else if (q < 145) v = reinterpret_cast<const float4 *>(corr + (long long)m * ld_corr)[q - 64];    // corr
    else if (q == 145) {
        const int rem = m % (h * w);
        const float fx = coords1[2 * (long long)m] - (float)(rem % w);
        const float fy = coords1[2 * (long long)m + 1] - (float)(rem / w);
        v = make_float4(fx, fy, delta[2 * (long long)m], delta[2 * (long long)m + 1]);
        flow_lr[2 * (long long)m] = fx;
        flow_lr[2 * (long long)m + 1] = fy;
    } else v = reinterpret_cast<const float4 *>(hx + (long long)m * 384 + 256)[q - 146];         // motion
    reinterpret_cast<float4 *>(ouin + (long long)m * 712)[q] = v;
}

// ---------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------
enum WeightSlot {
    W_CONVC1, B_CONVC1, W_CONVC2, B_CONVC2, W_CONVF1, B_CONVF1, W_CONVF2, B_CONVF2, W_CONV, B_CONV,
    // GRU gates: *_DYN = columns of [h | motion] (evaluated every iteration), *_INP = columns of the
    // context features `inp` (evaluated once per pair, with the bias)
    W_ZR1_DYN, W_ZR1_INP, B_ZR1, W_Q1_DYN, W_Q1_INP, B_Q1, W_ZR2_DYN, W_ZR2_INP, B_ZR2, W_Q2_DYN, W_Q2_INP, B_Q2,
    W_FH1, B_FH1, W_FH2, B_FH2, W_MASK0, B_MASK0, W_MASK2, B_MASK2,
    W_OU1, B_OU1, W_OU2, B_OU2, W_COUNT
};

struct Workspace {
    float *lvl[4];
    float *f2l[3];                 // on-demand correlation: the pooled second feature map, levels 1..3
    float *coords1, *corr, *cor1, *corflo, *flo1, *hx, *z, *rh, *fh, *delta, *mask, *ouin, *ouh, *ou, *flow_lr;
    float *pre_zr[2], *pre_q[2];   // inp part of the GRU gate convolutions (+ bias), per pass
    float *f2s;                    // split form of fmap2 (B operand of the volume GEMM in split arithmetic)
    float *hf;                     // split arithmetic: fp32 copy of h [M][128] (hx itself is in split form)
    float *hb, *hfb;               // split arithmetic: h between the two passes of the fused GRU kernel [M][128]: split form, fp32
    int ld_corr;                   // 324: the lookup's features stay fp32 (the lookup is HBM-bound; convc1 splits them in registers)
    size_t bytes;
};

static Workspace carve(void *base, int P, int h, int w, bool ondemand = false, bool split = false) {
    Workspace ws{};
    const size_t N = (size_t)h * w, M = (size_t)P * N;
    size_t off = 0;
    auto take = [&](size_t floats) {
        float *p = base ? reinterpret_cast<float *>(static_cast<char *>(base) + off) : nullptr;
        off += (floats * sizeof(float) + 255) & ~size_t(255);
        return p;
    };
    const PyramidLayout L = pyramid_layout(h, w);
    for (int l = 0; l < 4; ++l) ws.lvl[l] = take(ondemand ? 0 : M * (size_t)L.stride[l]);
    for (int l = 1; l < 4; ++l) ws.f2l[l - 1] = take(ondemand ? (size_t)P * (h >> l) * (w >> l) * 256 : 0);
    ws.coords1 = take(M * 2);
    ws.ld_corr = 324;
    ws.corr = take(M * ws.ld_corr);
    ws.cor1 = take(M * 256);
    ws.corflo = take(M * 256);
    ws.flo1 = take(M * 128);
    ws.hx = take(M * 384);
    ws.z = take(M * 128);
    ws.rh = take(M * 128);
    ws.fh = take(M * 256);
    ws.delta = take(M * 2);
    ws.mask = take(M * 576);
    ws.ouin = take(M * 712);
    ws.ouh = take(M * 256);
    ws.ou = take(M * 4);
    ws.flow_lr = take(M * 2);
    for (int pass = 0; pass < 2; ++pass) { ws.pre_zr[pass] = take(M * 256); ws.pre_q[pass] = take(M * 128); }
    ws.f2s = take(ondemand || !split ? 0 : M * 256);
    ws.hf = take(split ? M * 128 : 0);
    ws.hb = take(split ? M * 128 : 0);
    ws.hfb = take(split ? M * 128 : 0);
    ws.bytes = off;
    return ws;
}

}  // namespace mftx

using namespace mftx;

struct mftx_raft {
    uint32_t magic;
    int ondemand;                  // 1: on-demand correlation (no volume), see csrc/corr_ondemand.hip
    int arith;                     // MFTX_ARITH_*: arithmetic of the update block's matrix products
    const float *w[W_COUNT];
    const float *wg[W_COUNT];      // what the GEMM layers stream: w, or the split form of it (arith = MFTX_ARITH_SPLIT)
    // the motion encoder's flow branch (convf1 -> convf2) runs on a stream of its own, beside lookup -> convc1 -> convc2
    hipStream_t side;
    hipEvent_t ev_fork, ev_join;
    float *coords_trace;           // debug payload (mftx_raft_set_coords_trace): coords1 before every iteration and after the last, or null
    GraphCache *graphs;            // the refinement's launch sequence between its first and last kernels, captured per (shape, workspace, mode)
    const void *wfused;            // convc1's weights for the fused lookup + convc1 kernel (csrc/lookup_convc1.hip), or null
    const void *wflow;             // convf1's and convf2's weights for the fused flow-branch kernel (csrc/flow_branch.hip), or null
    const void *wproj;             // the flow head's last layer as the projection epilogue of its first (csrc/tile_conv.hip: TC_RELU_PROJ), or null
    const void *wt[W_COUNT];       // weight streams of the tile-resident conv kernel (csrc/tile_conv.hip) per slot, or null
    const void *wou, *wouproj;     // the occlusion + uncertainty heads as one tile-resident kernel (csrc/tile_conv.hip: ou_head_kernel), or null
    int opt[13];                   // MFTX_RAFT_OPT_*
    unsigned *nonfinite;           // device counter of non-finite output pixels (mftx_raft_set_nonfinite_counter), or null
};
// weights that go through the conv GEMM (the others feed VALU kernels and stay fp32)
static constexpr int GEMM_SLOTS[] = {W_CONVC1, W_CONVC2, W_CONVF2, W_CONV, W_ZR1_DYN, W_ZR1_INP, W_Q1_DYN, W_Q1_INP,
                                     W_ZR2_DYN, W_ZR2_INP, W_Q2_DYN, W_Q2_INP, W_FH1, W_MASK0, W_MASK2, W_OU1};
static constexpr uint32_t RAFT_MAGIC = 0x4d465458;  // "MFTX"

extern "C" int mftx_raft_create(const float *const *weights, int n_weights, mftx_raft **out) {
    if (!weights || !out) return fail(MFTX_E_ARG, "raft_create: null pointer");
    if (n_weights != W_COUNT) return fail(MFTX_E_ARG, "raft_create: expected %d weight tensors, got %d", (int)W_COUNT, n_weights);
    for (int i = 0; i < W_COUNT; ++i) {
        if (!weights[i]) return fail(MFTX_E_ARG, "raft_create: weight %d is null", i);
        if (!aligned16(weights[i])) return fail(MFTX_E_ALIGN, "raft_create: weight %d not 16-byte aligned", i);
    }
    mftx_raft *r = new (std::nothrow) mftx_raft;
    if (!r) return fail(MFTX_E_ARG, "raft_create: out of host memory");
    r->magic = RAFT_MAGIC;
    r->ondemand = 0;
    r->arith = MFTX_ARITH_F32;
    r->side = nullptr; r->ev_fork = nullptr; r->ev_join = nullptr;
    r->wfused = nullptr;
    r->wflow = nullptr;
    r->wproj = nullptr;
    r->wou = nullptr; r->wouproj = nullptr;
    for (int i = 0; i < W_COUNT; ++i) r->wt[i] = nullptr;
    r->coords_trace = nullptr;
    r->nonfinite = nullptr;
    r->graphs = new (std::nothrow) GraphCache;
    r->opt[MFTX_RAFT_OPT_FORK] = -1; r->opt[MFTX_RAFT_OPT_PRESPLIT] = 1; r->opt[MFTX_RAFT_OPT_GROUP] = 1; r->opt[MFTX_RAFT_OPT_FUSE_LOOKUP] = 1; r->opt[MFTX_RAFT_OPT_GRAPH] = 1; r->opt[MFTX_RAFT_OPT_FUSE_FLOW] = 1; r->opt[MFTX_RAFT_OPT_TILE_CONV] = 1; r->opt[MFTX_RAFT_OPT_FUSE_HEAD] = 1; r->opt[MFTX_RAFT_OPT_TILE_VOLUME] = 1; r->opt[MFTX_RAFT_OPT_FUSE_GRU] = 1; r->opt[MFTX_RAFT_OPT_TILE_CELLS] = 0; r->opt[MFTX_RAFT_OPT_FUSE_OU] = 1; r->opt[MFTX_RAFT_OPT_TILE_CONV2P] = 1;
    for (int i = 0; i < W_COUNT; ++i) r->w[i] = r->wg[i] = weights[i];
    *out = r;
    return 0;
}

extern "C" void mftx_raft_destroy(mftx_raft *r) {
    if (r && r->magic == RAFT_MAGIC) {
        r->magic = 0;
        if (r->ev_fork) (void)hipEventDestroy(r->ev_fork);
        if (r->ev_join) (void)hipEventDestroy(r->ev_join);
        if (r->side) (void)hipStreamDestroy(r->side);
        delete r->graphs;
        delete r;
    }
}

// the side stream and its two events, on first use (on the device that is current in the calling thread)
static int ensure_side_stream(mftx_raft *r) {
    if (r->side) return 0;
    hipError_t e = hipStreamCreateWithFlags(&r->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r->ev_join, hipEventDisableTiming);
    if (e != hipSuccess) return fail((int)e, "raft_refine: side stream: %s", hipGetErrorString(e));
    return 0;
}

extern "C" size_t mftx_raft_workspace_bytes(int P, int h, int w) {
    if (P <= 0 || h <= 0 || w <= 0) return 0;
    return carve(nullptr, P, h, w).bytes;
}

extern "C" int mftx_raft_set_ondemand(mftx_raft *r, int on) {
    if (!r || r->magic != RAFT_MAGIC) return fail(MFTX_E_STATE, "raft_set_ondemand: bad handle");
    r->ondemand = on ? 1 : 0;
    if (r->graphs) r->graphs->clear();
    return 0;
}

extern "C" int mftx_raft_set_split_weights(mftx_raft *r, const void *const *split, int n) {
    if (!r || r->magic != RAFT_MAGIC) return fail(MFTX_E_STATE, "raft_set_split_weights: bad handle");
    if (r->graphs) r->graphs->clear();
    if (!split) {                                    // back to fp32 MFMA
        for (int i = 0; i < W_COUNT; ++i) r->wg[i] = r->w[i];
        r->arith = MFTX_ARITH_F32;
        return 0;
    }
    if (n != W_COUNT) return fail(MFTX_E_ARG, "raft_set_split_weights: expected %d slots, got %d", (int)W_COUNT, n);
    for (int slot : GEMM_SLOTS) {
        if (!split[slot]) return fail(MFTX_E_ARG, "raft_set_split_weights: slot %d (a GEMM layer) is null", slot);
        if (!aligned16(split[slot])) return fail(MFTX_E_ALIGN, "raft_set_split_weights: slot %d not 16-byte aligned", slot);
    }
    for (int slot : GEMM_SLOTS) r->wg[slot] = static_cast<const float *>(split[slot]);
    r->arith = MFTX_ARITH_SPLIT;
    return 0;
}

extern "C" int mftx_raft_set_lookup_fused(mftx_raft *r, const void *wfused) {
    if (!r || r->magic != RAFT_MAGIC) return fail(MFTX_E_STATE, "raft_set_lookup_fused: bad handle");
    if (wfused && !aligned16(wfused)) return fail(MFTX_E_ALIGN, "raft_set_lookup_fused: weights not 16-byte aligned");
    r->wfused = wfused;
    if (r->graphs) r->graphs->clear();
    return 0;
}

extern "C" int mftx_raft_set_flow_fused(mftx_raft *r, const void *wflow) {
    if (!r || r->magic != RAFT_MAGIC) return fail(MFTX_E_STATE, "raft_set_flow_fused: bad handle");
    if (wflow && !aligned16(wflow)) return fail(MFTX_E_ALIGN, "raft_set_flow_fused: weights not 16-byte aligned");
    r->wflow = wflow;
    if (r->graphs) r->graphs->clear();
    return 0;
}

extern "C" int mftx_raft_set_tile_weights(mftx_raft *r, const void *const *tile, int n) {
    if (!r || r->magic != RAFT_MAGIC) return fail(MFTX_E_STATE, "raft_set_tile_weights: bad handle");
    if (r->graphs) r->graphs->clear();
    if (!tile) { for (int i = 0; i < W_COUNT; ++i) r->wt[i] = nullptr; return 0; }
    if (n != W_COUNT) return fail(MFTX_E_ARG, "raft_set_tile_weights: expected %d slots, got %d", (int)W_COUNT, n);
    for (int i = 0; i < W_COUNT; ++i)
        if (tile[i] && !aligned16(tile[i])) return fail(MFTX_E_ALIGN, "raft_set_tile_weights: slot %d not 16-byte aligned", i);
    for (int i = 0; i < W_COUNT; ++i) r->wt[i] = tile[i];
    return 0;
}

extern "C" int mftx_raft_set_flow_head(mftx_raft *r, const void *wproj) {
    if (!r || r->magic != RAFT_MAGIC) return fail(MFTX_E_STATE, "raft_set_flow_head: bad handle");
    if (wproj && !aligned16(wproj)) return fail(MFTX_E_ALIGN, "raft_set_flow_head: weights not 16-byte aligned");
    r->wproj = wproj;
    if (r->graphs) r->graphs->clear();
    return 0;
}

extern "C" int mftx_raft_set_ou_heads(mftx_raft *r, const void *wtile, const void *wproj) {
    if (!r || r->magic != RAFT_MAGIC) return fail(MFTX_E_STATE, "raft_set_ou_heads: bad handle");
    if ((wtile == nullptr) != (wproj == nullptr)) return fail(MFTX_E_ARG, "raft_set_ou_heads: both weight streams, or neither");
    if (wtile && (!aligned16(wtile) || !aligned16(wproj))) return fail(MFTX_E_ALIGN, "raft_set_ou_heads: weights not 16-byte aligned");
    r->wou = wtile; r->wouproj = wproj;
    if (r->graphs) r->graphs->clear();
    return 0;
}

extern "C" int mftx_raft_set_coords_trace(mftx_raft *r, float *trace) {
    if (!r || r->magic != RAFT_MAGIC) return fail(MFTX_E_STATE, "raft_set_coords_trace: bad handle");
    r->coords_trace = trace;
    return 0;
}

// graphs are keyed by workspace address: a caller that frees or replaces a workspace drops the graphs captured on it
extern "C" int mftx_raft_clear_graphs(mftx_raft *r) {
    if (!r || r->magic != RAFT_MAGIC) return fail(MFTX_E_STATE, "raft_clear_graphs: bad handle");
    if (r->graphs) r->graphs->clear();
    return 0;
}

extern "C" int mftx_raft_set_nonfinite_counter(mftx_raft *r, unsigned *counter) {
    if (!r || r->magic != RAFT_MAGIC) return fail(MFTX_E_STATE, "raft_set_nonfinite_counter: bad handle");
    if (counter && (reinterpret_cast<uintptr_t>(counter) & 3)) return fail(MFTX_E_ALIGN, "raft_set_nonfinite_counter: counter not 4-byte aligned");
    r->nonfinite = counter;        // (read by the last kernel of a refinement, which is not part of the captured graph)
    return 0;
}

extern "C" int mftx_raft_set_option(mftx_raft *r, int option, int value) {
    if (!r || r->magic != RAFT_MAGIC) return fail(MFTX_E_STATE, "raft_set_option: bad handle");
    if (option < 0 || option > MFTX_RAFT_OPT_TILE_CONV2P) return fail(MFTX_E_ARG, "raft_set_option: unknown option %d", option);
    r->opt[option] = value;
    if (r->graphs) r->graphs->clear();
    return 0;
}

extern "C" int mftx_raft_arith(const mftx_raft *r) {
    return (r && r->magic == RAFT_MAGIC) ? r->arith : -1;
}

extern "C" size_t mftx_raft_workspace_bytes_for(const mftx_raft *r, int P, int h, int w) {
    if (!r || r->magic != RAFT_MAGIC || P <= 0 || h <= 0 || w <= 0) return 0;
    return carve(nullptr, P, h, w, r->ondemand != 0, r->arith == MFTX_ARITH_SPLIT).bytes;
}

// Byte offsets of the workspace regions, in the order lvl0..3, coords1, corr,
// cor1, corflo, flo1, hx, z, rh, fh, delta, mask, ouin, ouh, ou, flow_lr -- lets
// the parity tests inspect the intermediates of the last iteration.
static int workspace_layout(const mftx_raft *r, int P, int h, int w, size_t *offsets, int n) {
    if (P <= 0 || h <= 0 || w <= 0 || !offsets || n != 19) return fail(MFTX_E_ARG, "workspace_layout: need 19 slots");
    char *base = reinterpret_cast<char *>(uintptr_t(1) << 40);
    Workspace ws = carve(base, P, h, w, r && r->ondemand != 0, r && r->arith == MFTX_ARITH_SPLIT);
    float *ptrs[19] = {ws.lvl[0], ws.lvl[1], ws.lvl[2], ws.lvl[3], ws.coords1, ws.corr, ws.cor1, ws.corflo, ws.flo1,
                       ws.hx, ws.z, ws.rh, ws.fh, ws.delta, ws.mask, ws.ouin, ws.ouh, ws.ou, ws.flow_lr};
    for (int i = 0; i < 19; ++i) offsets[i] = (size_t)(reinterpret_cast<char *>(ptrs[i]) - base);
    return 0;
}
extern "C" int mftx_raft_workspace_layout(int P, int h, int w, size_t *offsets, int n) {
    return workspace_layout(nullptr, P, h, w, offsets, n);
}
extern "C" int mftx_raft_workspace_layout_for(const mftx_raft *r, int P, int h, int w, size_t *offsets, int n) {
    if (!r || r->magic != RAFT_MAGIC) return fail(MFTX_E_STATE, "workspace_layout_for: bad handle");
    return workspace_layout(r, P, h, w, offsets, n);
}

#define TRY(expr) do { int _e = (expr); if (_e) return _e; } while (0)

static mftx_conv_desc conv_desc(const float *a0, int lda0, int c0, const float *a1, int lda1, int c1,
                                const float *w, const float *b, float *out, int ldo, int P, int h, int wd, int N,
                                int kh, int kw, int act, float scale = 1.f, const float *addend = nullptr,
                                int ld_addend = 0) {
    mftx_conv_desc d{};
    d.addend = addend; d.ld_addend = ld_addend;
    d.a0 = a0; d.lda0 = lda0; d.c0 = c0; d.a1 = a1; d.lda1 = lda1; d.c1 = c1;
    d.wpk = w; d.bias = b; d.out = out; d.ldo = ldo; d.P = P; d.h = h; d.w = wd; d.N = N;
    d.kh = kh; d.kw = kw; d.act = act; d.out_scale = scale;
    return d;
}

// gather (optional): the pairs' maps through per-pair pointers (fmap1 / fmap2 / net / inp are then unused); f2_shared: every
// pair has the SAME second feature map (gather->f2.p[0])
struct RefineGather { PairPtrs f1, f2, net, inp; bool f2_shared; };
static int refine_impl(mftx_raft *r, int P, int h, int w, int iters, const float *fmap1,
                                const float *fmap2, const float *net, const float *inp, const float *flow_init,
                                int pad_left,
                                int pad_right, int pad_top, int pad_bottom, float *flow, float *occl,
                                float *sigma, float *packed, float *flow_lr_out, void *workspace,
                                size_t workspace_bytes, void *stream, const RefineGather *gather) {
    if (!r || r->magic != RAFT_MAGIC) return fail(MFTX_E_STATE, "raft_refine: bad handle");
    const bool planar = flow && occl && sigma;
    if (gather) { fmap1 = gather->f1.p[0]; fmap2 = gather->f2.p[0]; net = gather->net.p[0]; inp = gather->inp.p[0]; }      // (for the checks below)
    if (!fmap1 || !fmap2 || !net || !inp || !workspace || (!planar && (flow || occl || sigma || !packed)))
        return fail(MFTX_E_ARG, "raft_refine: null pointer (outputs: flow + occl + sigma, or packed, or both)");
    if (P <= 0 || h < 16 || w < 16 || iters < 1)
        return fail(MFTX_E_ARG, "raft_refine: need P >= 1, h, w >= 16 (level 3 of the pyramid needs >= 2 cells), iters >= 1");
    if ((long long)P * h * w > (1ll << 24)) return fail(MFTX_E_ARG, "raft_refine: batch too large");
    if (!aligned16(fmap1) || !aligned16(fmap2) || !aligned16(net) || !aligned16(inp) ||
        (reinterpret_cast<uintptr_t>(workspace) & 255))
        return fail(MFTX_E_ALIGN, "raft_refine: inputs must be 16-byte and the workspace 256-byte aligned");
    if (pad_left < 0 || pad_right < 0 || pad_top < 0 || pad_bottom < 0 || pad_left + pad_right >= 8 ||
        pad_top + pad_bottom >= 8)
        return fail(MFTX_E_ARG, "raft_refine: bad padding");
    const bool ondemand = r->ondemand != 0;
    // split arithmetic: every tensor that feeds a GEMM of the update block / OU heads lives in the workspace in SPLIT
    // form (common.h), written that way by its producer -- the GEMMs' K loops then spend nothing on splitting
    const bool SP = r->arith == MFTX_ARITH_SPLIT && r->opt[MFTX_RAFT_OPT_PRESPLIT] != 0;
    // lookup fused into convc1 (csrc/lookup_convc1.hip): the 324 features stay in LDS; they are materialised on the last
    // iteration only, for the occlusion / uncertainty heads
    const bool fuse_lookup = SP && !ondemand && r->wfused != nullptr && r->opt[MFTX_RAFT_OPT_FUSE_LOOKUP] != 0 &&
                             lookup_convc1_applicable(P, h, w, 256);
    Workspace ws = carve(workspace, P, h, w, ondemand, r->arith == MFTX_ARITH_SPLIT);
    if (ws.bytes > workspace_bytes)
        return fail(MFTX_E_WORKSPACE, "raft_refine: workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
    hipStream_t s = (hipStream_t)stream;
    const bool use_graph = r->graphs && r->opt[MFTX_RAFT_OPT_GRAPH] != 0 && !prof_enabled() && !r->coords_trace && !ondemand;
    hipStream_t caller = s;
    bool proxied = false;
    if (use_graph && s == nullptr) {                 // the legacy stream (PyTorch's default) cannot be captured: graph_cache.h
        hipStream_t own = r->graphs->proxy.enter(s);
        if (own) { s = own; proxied = true; }
    }
    struct Leave { GraphCache *g; hipStream_t caller; bool on; ~Leave() { if (on) g->proxy.leave(caller); } } leave{r->graphs, caller, proxied};
    const int N = h * w, M = P * N;
    const float *const *W = r->w;
    const float *const *G = r->wg;          // GEMM layers: fp32 or split weights, by the handle's arithmetic
    const int AR = r->arith;
    // gemm(desc, a, o): arithmetic of the handle; a: the A operand(s) are in split form, o: the output is written in it
    auto gemm = [AR, SP](mftx_conv_desc d, bool a = true, bool o = false) { d.arith = AR; d.a_split = SP && a; d.out_split = SP && o; return d; };

    // correlation volume + pyramid (core/corr.py:14-28)
    const float *f2lv[4] = {fmap2, ws.f2l[0], ws.f2l[1], ws.f2l[2]};
    if (ondemand) TRY(launch_fmap_pyramid(fmap2, P, 256, h, w, ws.f2l, s));   // core/corr.py:78-82 (only fmap2's pyramid is used)
    else if (gather) {
        // gathered pairs (split arithmetic, tile-resident volume: checked by the caller): the second maps are split into the
        // workspace -- once when all pairs share one -- and the volume kernel takes every pair's first map where it lies
        const long long pair_floats = (long long)N * 256;
        if (gather->f2_shared) TRY(launch_split_weights(gather->f2.p[0], ws.f2s, pair_floats, s));
        else for (int b = 0; b < P; ++b) TRY(launch_split_weights(gather->f2.p[b], ws.f2s + b * pair_floats, pair_floats, s));
        TRY(launch_volume_tile(nullptr, ws.f2s, P, h, w, ws.lvl, s, &gather->f1, gather->f2_shared ? 0 : pair_floats));
    }
    else TRY(launch_corr_pyramid(fmap1, fmap2, P, 256, h, w, ws.lvl, s, r->arith == MFTX_ARITH_SPLIT ? ws.f2s : nullptr, r->opt[MFTX_RAFT_OPT_TILE_VOLUME]));
    {
        const long long slots = (long long)M * 64;
        ProfScope prof(PC_GLUE, s, 0);
        InitGather ga{};
        if (gather) { ga.on = 1; ga.netp = gather->net; ga.inpp = gather->inp; }
        hipLaunchKernelGGL(init_state_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, s, net, inp,
                           flow_init, ws.hx, SP ? ws.hf : nullptr, ws.coords1, M, h, w, ga);
        TRY(check_launch("init_state"));
    }
    float *flow_lr = flow_lr_out ? flow_lr_out : ws.flow_lr;
    // Everything between the first kernels (which read the caller's feature maps) and the last (which writes the caller's
    // outputs) touches the workspace only: one launch sequence per (shape, workspace, mode), replayed as a hipGraph from
    // its third use on (graph_cache.h).  The forked side stream joins the capture through its events.
    auto core = [&]() -> int {
    // The gate convolutions are linear in their input [h | inp | motion] and `inp` does not change
    // over the iterations (core/raft.py:146-149): its third of every gate sum (+ bias) is computed
    // once here and enters the per-iteration GEMMs as an epilogue addend.  Same terms, summed once.
    // Layers whose input tile fits a CU's LDS run on the tile-resident kernel (csrc/tile_conv.hip) when its weight streams are set
    // (option value 2: whatever the batch; 1, the default: only when the tiles fill the chip -- at 256 x 256 pixels or one pair
    // per GPU the ring-buffered kernel's small tiles win)
    const bool tiles_on = SP && r->opt[MFTX_RAFT_OPT_TILE_CONV] != 0 &&
                          (r->opt[MFTX_RAFT_OPT_TILE_CONV] == 2 ||
                           (tile_conv_fills_chip(P, h, w, 3, 3) && tile_conv_fills_chip(P, h, w, 1, 5) && tile_conv_fills_chip(P, h, w, 5, 1)));
    auto tile_w = [&](int slot) -> const void * { return tiles_on ? r->wt[slot] : nullptr; };
    auto tile_layer = [&](const float *a0, int lda0, const float *a1, int lda1, const void *wf, const float *bias, int N, int kh, int kw, int epi) {
        TileConvLaunch t{};
        t.a0 = a0; t.lda0 = lda0; t.a1 = a1; t.lda1 = lda1; t.cin = a1 ? 256 : 128; t.wf = wf; t.bias = bias;
        t.P = P; t.h = h; t.w = w; t.N = N; t.kh = kh; t.kw = kw; t.epi = epi;
        t.cells = r->opt[MFTX_RAFT_OPT_TILE_CELLS];
        return t;
    };
    for (int pass = 0; pass < 2; ++pass) {
        const int kh = pass ? 5 : 1, kw = pass ? 1 : 5;
        const int szr = pass ? W_ZR2_INP : W_ZR1_INP, sq = pass ? W_Q2_INP : W_Q1_INP;
        if (tile_w(szr)) {
            TileConvLaunch t = tile_layer(ws.hx + 128, 384, nullptr, 0, tile_w(szr), W[pass ? B_ZR2 : B_ZR1], 256, kh, kw, 0);
            t.out = ws.pre_zr[pass]; t.ldo = 256;
            TRY(launch_tile_conv(t, s));
        } else TRY(launch_conv(gemm(conv_desc(ws.hx + 128, 384, 128, nullptr, 0, 0, G[szr], W[pass ? B_ZR2 : B_ZR1], ws.pre_zr[pass], 256, P, h, w, 256, kh, kw, 0), true, false), s));
        if (tile_w(sq)) {
            TileConvLaunch t = tile_layer(ws.hx + 128, 384, nullptr, 0, tile_w(sq), W[pass ? B_Q2 : B_Q1], 128, kh, kw, 0);
            t.out = ws.pre_q[pass]; t.ldo = 128;
            TRY(launch_tile_conv(t, s));
        } else TRY(launch_conv(gemm(conv_desc(ws.hx + 128, 384, 128, nullptr, 0, 0, G[sq], W[pass ? B_Q2 : B_Q1], ws.pre_q[pass], 128, P, h, w, 128, kh, kw, 0), true, false), s));
    }
    const float *lv[4] = {ws.lvl[0], ws.lvl[1], ws.lvl[2], ws.lvl[3]};
    const int strips = cdiv(w, F1_CELLS);
    // The coordinates the iteration works on.  With the flow branch and the flow head both fused, an iteration's update is
    // applied by the NEXT iteration's flow-branch kernel (which reads every cell of its tile anyway) into the other of two
    // buffers -- flo1 is free then -- instead of by a launch of its own; only the last update is applied by
    // flow_head_sum_kernel, into coords1.
    float *ccur = ws.coords1, *calt = ws.flo1;
    bool pending = false;                    // ws.fh holds an update (T) that ccur does not contain yet
    for (int it = 0; it < iters; ++it) {
        const bool last = (it == iters - 1);
        if (r->coords_trace &&       // RAFT.forward(vis_debug=True): the coordinates every iteration starts from (core/raft.py:175-176)
            hipMemcpyAsync(r->coords_trace + (size_t)it * M * 2, ccur, (size_t)M * 8, hipMemcpyDeviceToDevice, s) != hipSuccess)
            return fail(MFTX_E_STATE, "raft_refine: coords trace copy failed");
        // The motion encoder has two independent branches (core/update.py:152-158): correlation lookup -> convc1 -> convc2
        // and convf1 -> convf2 on the flow.  Both need only coords1.  Split arithmetic, small batches (up to four
        // 512 x 512 pairs: the kernels leave CUs idle): the flow branch runs on the handle's SIDE STREAM beside the
        // correlation branch and joins in front of `conv` -- 3.67 -> 3.45 ms at one pair, 5.94 -> 5.59 at four
        // (tools/bench_pairs.py).  At seven pairs this used to lose (every kernel filled the chip: 106.7 vs 111.5 frames/s);
        // since convc2 runs as 224 workgroups of the 128 x 192 tile, 32 CUs are free beside it and its 1 x 1 predecessor
        // for the flow branch's small kernels: 129.2 -> 130.3 frames/s (MFTX_RAFT_OPT_FORK = 0: in order on one stream, with
        // lookup + convf1 as one launch).  fp32 MFMA keeps round 1's grouping (lookup + convf1 in one launch,
        // convc2 + convf2 in one launch).  The per-kernel timing pass and MFTX_RAFT_OPT_GROUP = 0 run everything in order.
        const ConvF1Args f1{ccur, W[W_CONVF1], W[B_CONVF1], ws.flo1, ws.hx, h, w, strips, P * h * strips, SP ? 1 : 0};
        const int f1_blocks = cdiv(f1.n_strips, 2);
        const bool nofuse = r->opt[MFTX_RAFT_OPT_GROUP] == 0, nopair = nofuse;
        const mftx_conv_desc c2 = gemm(conv_desc(ws.cor1, 256, 256, nullptr, 0, 0, G[W_CONVC2], W[B_CONVC2], ws.corflo, 256, P, h, w, 192, 3, 3, 1), true, true);
        const mftx_conv_desc f2 = gemm(conv_desc(ws.flo1, 128, 128, nullptr, 0, 0, G[W_CONVF2], W[B_CONVF2], ws.corflo + 192, 256, P, h, w, 64, 3, 3, 1), true, true);
        const int fork_env = r->opt[MFTX_RAFT_OPT_FORK];      // 0 never, -1 / 1 with the split arithmetic
        // the flow branch as ONE kernel (csrc/flow_branch.hip), in order on this stream: no side stream, no join
        const bool fuse_flow = SP && r->wflow != nullptr && r->opt[MFTX_RAFT_OPT_FUSE_FLOW] != 0 && !nofuse;
        if (fuse_flow) {
            if (pending) {
                TRY(launch_flow_branch(ccur, P, h, w, r->wflow, W[B_CONVF1], W[B_CONVF2], ws.corflo + 192, 256, ws.hx, 384, s, ws.fh, W[B_FH2], calt, ws.delta));
                float *t = ccur; ccur = calt; calt = t;
                pending = false;
            } else TRY(launch_flow_branch(ccur, P, h, w, r->wflow, W[B_CONVF1], W[B_CONVF2], ws.corflo + 192, 256, ws.hx, 384, s));
        }
        const bool serial = fuse_flow || prof_enabled() || nofuse || (AR == MFTX_ARITH_SPLIT && (fork_env == 0 || fork_env == 2));
        const bool forked = !serial && AR == MFTX_ARITH_SPLIT;
        const bool flow_first = !fuse_flow && serial && fuse_lookup && fork_env == 2 && !prof_enabled() && !nofuse;    // convf1, convf2, then the correlation branch
        if (flow_first) {
            hipLaunchKernelGGL(convf1_kernel, dim3(f1_blocks), dim3(256), 0, s, f1);
            TRY(check_launch("convf1"));
            TRY(launch_conv(f2, s));
        }
        if (forked) {
            TRY(ensure_side_stream(r));
            if (hipEventRecord(r->ev_fork, s) != hipSuccess || hipStreamWaitEvent(r->side, r->ev_fork, 0) != hipSuccess)
                return fail(MFTX_E_STATE, "raft_refine: fork onto the side stream failed");
            hipLaunchKernelGGL(convf1_kernel, dim3(f1_blocks), dim3(256), 0, r->side, f1);
            TRY(check_launch("convf1"));
            TRY(launch_conv(f2, r->side));
            // fused lookup: the features themselves are needed once, by ou_gather behind the last iteration -- off the critical path
            if (fuse_lookup && last) TRY(launch_corr_lookup(lv, ccur, P, h, w, ws.corr, ws.ld_corr, r->side));
            if (hipEventRecord(r->ev_join, r->side) != hipSuccess) return fail(MFTX_E_STATE, "raft_refine: join event failed");
        }
        // lookup + convf1 as ONE launch (HBM gathers beside VALU work) whenever the flow branch is not on the side stream:
        // 113.7 vs 113.2 frames/s with the split arithmetic; the timing pass and MFTX_RAFT_OPT_GROUP = 0 keep them apart
        if (fuse_lookup) {
            if (last && !forked) TRY(launch_corr_lookup(lv, ccur, P, h, w, ws.corr, ws.ld_corr, s));
            TRY(launch_lookup_convc1(lv, ccur, P, h, w, r->wfused, W[B_CONVC1], ws.cor1, 256, 1, s));
            if (!forked && !flow_first && !fuse_flow) {
                ProfScope prof(PC_CONVF1, s, 2.0 * M * 128 * 98);
                hipLaunchKernelGGL(convf1_kernel, dim3(f1_blocks), dim3(256), 0, s, f1);
            }
        } else if (prof_enabled() || nofuse || forked || ondemand || fuse_flow) {
            // (the 324 features stay fp32: written in split form the lookup takes 31 instead of 27 us, more than convc1
            // gains from a pre-split A -- and its HBM roofline is the one with a north-star target)
            if (ondemand) TRY(launch_corr_ondemand(fmap1, f2lv, ccur, P, h, w, ws.corr, ws.ld_corr, s));
            else TRY(launch_corr_lookup(lv, ccur, P, h, w, ws.corr, ws.ld_corr, s));
            if (!forked && !fuse_flow) {
                ProfScope prof(PC_CONVF1, s, 2.0 * M * 128 * 98);
                hipLaunchKernelGGL(convf1_kernel, dim3(f1_blocks), dim3(256), 0, s, f1);
            }
        } else {
            const LookupArgs la = make_lookup_args(lv, ccur, P, h, w, ws.corr, ws.ld_corr);
            const int lookup_blocks = cdiv(cdiv(la.cells, 2), LK_WAVES);
            hipLaunchKernelGGL(lookup_convf1_kernel, dim3(lookup_blocks + f1_blocks), dim3(256), 0, s, la, f1,
                               lookup_blocks, f1_blocks);
        }
        TRY(check_launch("lookup + convf1"));
        // motion encoder (core/update.py:152-160)
        if (!fuse_lookup) TRY(launch_conv(gemm(conv_desc(ws.corr, ws.ld_corr, ws.ld_corr, nullptr, 0, 0, G[W_CONVC1], W[B_CONVC1], ws.cor1, 256, P, h, w, 256, 1, 1, 1), false, true), s));
        // convc2 and conv (3 x 3 over 256 channels): tile-resident in two channel passes where the tile-resident layers run (round 6)
        const bool two_pass = tiles_on && r->opt[MFTX_RAFT_OPT_TILE_CONV2P] != 0 && r->wt[W_CONVC2] && r->wt[W_CONV];
        auto run_c2 = [&]() -> int {
            if (two_pass) return launch_tile_conv2p(ws.cor1, 256, r->wt[W_CONVC2], W[B_CONVC2], ws.corflo, 256, 192, P, h, w, r->opt[MFTX_RAFT_OPT_TILE_CELLS], s);
            return launch_conv(c2, s);
        };
        if (forked) {
            TRY(run_c2());
            if (hipStreamWaitEvent(s, r->ev_join, 0) != hipSuccess) return fail(MFTX_E_STATE, "raft_refine: join failed");
        } else if (nopair || AR != MFTX_ARITH_F32) {
            TRY(run_c2());
            if (!flow_first && !fuse_flow) TRY(launch_conv(f2, s));
        } else {
            TRY(launch_conv_pair(c2, f2, s));      // second layers of the two branches in one launch
        }
        if (two_pass) TRY(launch_tile_conv2p(ws.corflo, 256, r->wt[W_CONV], W[B_CONV], ws.hx + 256, 384, 126, P, h, w, r->opt[MFTX_RAFT_OPT_TILE_CELLS], s));
        else TRY(launch_conv(gemm(conv_desc(ws.corflo, 256, 256, nullptr, 0, 0, G[W_CONV], W[B_CONV], ws.hx + 256, 384, P, h, w, 126, 3, 3, 1), true, true), s));
        // SepConvGRU (core/update.py:108-123): horizontal 1x5 then vertical 5x1
        // (decided ONCE for both passes: they hand h over through the ping-pong pair hx/hf <-> hb/hfb, so one fused and one
        // unfused pass would read a buffer the other never wrote -- a partial set of tile weights runs both passes unfused)
        const bool gru_fused = tile_w(W_ZR1_DYN) && tile_w(W_Q1_DYN) && tile_w(W_ZR2_DYN) && tile_w(W_Q2_DYN) &&
                               r->opt[MFTX_RAFT_OPT_FUSE_GRU] != 0;
        for (int pass = 0; pass < 2; ++pass) {
            const int kh = pass ? 5 : 1, kw = pass ? 1 : 5;
            const int szr = pass ? W_ZR2_DYN : W_ZR1_DYN, sq = pass ? W_Q2_DYN : W_Q1_DYN;
            if (gru_fused) {
                // the whole pass as ONE kernel (tile_conv.hip: gru_half_kernel): the tile is loaded once, r * h stays in LDS; h goes
                // hx -> hb in the horizontal pass and back in the vertical one (a tile's halo cells are its neighbours' outputs)
                GruHalfLaunch g{};
                g.h_in = pass ? ws.hb : ws.hx; g.ld_hin = pass ? 128 : 384; g.h_out = pass ? ws.hx : ws.hb; g.ld_hout = pass ? 384 : 128;
                g.mo = ws.hx + 256; g.ld_mo = 384; g.wzr = tile_w(szr); g.wq = tile_w(sq); g.pre_zr = ws.pre_zr[pass]; g.pre_q = ws.pre_q[pass];
                g.z = ws.z; g.hf_in = pass ? ws.hfb : ws.hf; g.hf_out = pass ? ws.hf : ws.hfb; g.P = P; g.h = h; g.w = w; g.pass = pass; g.cells = r->opt[MFTX_RAFT_OPT_TILE_CELLS];
                TRY(launch_gru_half(g, s));
                continue;
            }
            if (tile_w(szr)) {
                TileConvLaunch t = tile_layer(ws.hx, 384, ws.hx + 256, 384, tile_w(szr), nullptr, 256, kh, kw, 2);
                t.addend = ws.pre_zr[pass]; t.ld_addend = 256; t.z = ws.z; t.rh = ws.rh; t.hf = ws.hf; t.ld_hf = 128;
                TRY(launch_tile_conv(t, s));
            } else {
                GruEpilogue g1{1, ws.hx, 384, ws.z, ws.rh, SP ? ws.hf : nullptr, 128};
                TRY(launch_conv_gru(gemm(conv_desc(ws.hx, 384, 128, ws.hx + 256, 384, 128, G[szr], nullptr, ws.z, 128, P, h, w, 256, kh, kw, 2, 1.f, ws.pre_zr[pass], 256), true, true), g1, s));
            }
            if (tile_w(sq)) {
                TileConvLaunch t = tile_layer(ws.rh, 128, ws.hx + 256, 384, tile_w(sq), nullptr, 128, kh, kw, 3);
                t.addend = ws.pre_q[pass]; t.ld_addend = 128; t.z = ws.z; t.hf = ws.hf; t.ld_hf = 128; t.hx = ws.hx; t.ld_hx = 384;
                TRY(launch_tile_conv(t, s));
            } else {
                GruEpilogue g2{2, ws.hx, 384, ws.z, ws.rh, SP ? ws.hf : nullptr, 128};
                TRY(launch_conv_gru(gemm(conv_desc(ws.rh, 128, 128, ws.hx + 256, 384, 128, G[sq], nullptr, ws.hx, 384, P, h, w, 128, kh, kw, 3, 1.f, ws.pre_q[pass], 128), true, true), g2, s));
            }
        }
        // flow head (core/update.py:6-14) and coordinate update (core/raft.py:184)
        const bool head_fused = tile_w(W_FH1) && r->wproj != nullptr && r->opt[MFTX_RAFT_OPT_FUSE_HEAD] != 0;
        if (head_fused) {
            // both layers of the flow head: relu(conv1) stays in LDS, multiplied there with conv2's filter as [256 x 18] partial
            // products per cell (-> ws.fh, [M][18]); the nine shifted terms are added, and coords1 updated, by a small kernel
            TileConvLaunch t = tile_layer(ws.hx, 384, nullptr, 0, tile_w(W_FH1), W[B_FH1], 256, 3, 3, 4);
            t.wproj = r->wproj; t.tout = ws.fh;
            TRY(launch_tile_conv(t, s));
            // (the update is left pending for the next iteration's flow-branch kernel when that kernel runs; the last one, and every
            // one under a debug trace, is applied here -- the last into coords1, whichever buffer is current)
            const bool defer = !last && fuse_flow && r->opt[MFTX_RAFT_OPT_FUSE_HEAD] != 2 && !r->coords_trace;       // (fuse_flow: the same for every iteration)
            if (defer) pending = true;
            else {
                float *dst = last ? ws.coords1 : ccur;
                TRY(launch_flow_head_sum(ws.fh, W[B_FH2], ws.delta, ccur, dst, P, h, w, s));
                ccur = dst;
            }
        } else if (tile_w(W_FH1)) {
            TileConvLaunch t = tile_layer(ws.hx, 384, nullptr, 0, tile_w(W_FH1), W[B_FH1], 256, 3, 3, 1);
            t.out = ws.fh; t.ldo = 256;
            TRY(launch_tile_conv(t, s));
        } else TRY(launch_conv(gemm(conv_desc(ws.hx, 384, 128, nullptr, 0, 0, G[W_FH1], W[B_FH1], ws.fh, 256, P, h, w, 256, 3, 3, 1), true, false), s));
        // last layer of the flow head, fused with coords1 += delta_flow (core/raft.py:184)
        if (!head_fused) {
            const mftx_conv_desc fh2 = conv_desc(ws.fh, 256, 256, nullptr, 0, 0, W[W_FH2], W[B_FH2], ws.delta, 2, P, h, w, 2, 3, 3, 0);
            if (!conv_small_applicable(fh2)) return fail(MFTX_E_STATE, "raft_refine: flow-head layer does not fit the small-N kernel");
            TRY(launch_conv_small(fh2, s, ccur, 2));
        }
        if (!last) continue;
        if (r->coords_trace &&       // ... and the final ones (core/raft.py:255-256)
            hipMemcpyAsync(r->coords_trace + (size_t)iters * M * 2, ccur, (size_t)M * 8, hipMemcpyDeviceToDevice, s) != hipSuccess)
            return fail(MFTX_E_STATE, "raft_refine: coords trace copy failed");
        // The upsampling mask is consumed only after the last iteration in test
        // mode (core/raft.py:190-196,234-239), so it is computed once.
        // (the hidden 256 channels go to the 1 x 1 layer in split form: a GEMM that splits its A operand in registers runs at half the
        // matrix utilisation of one that finds it split -- profiles/r4k_pmc_mfma_util.csv: 0.18 against 0.35)
        if (tile_w(W_MASK0)) {
            TileConvLaunch t = tile_layer(ws.hx, 384, nullptr, 0, tile_w(W_MASK0), W[B_MASK0], 256, 3, 3, 1);
            t.out = ws.fh; t.ldo = 256; t.out_split = SP ? 1 : 0;
            TRY(launch_tile_conv(t, s));
        } else TRY(launch_conv(gemm(conv_desc(ws.hx, 384, 128, nullptr, 0, 0, G[W_MASK0], W[B_MASK0], ws.fh, 256, P, h, w, 256, 3, 3, 1), true, true), s));
        TRY(launch_conv(gemm(conv_desc(ws.fh, 256, 256, nullptr, 0, 0, G[W_MASK2], W[B_MASK2], ws.mask, 576, P, h, w, 576, 1, 1, 0, 0.25f), true, false), s));
        // occlusion + uncertainty heads (core/update.py:196-214)
        const bool ou_fused = tiles_on && r->wou != nullptr && r->opt[MFTX_RAFT_OPT_FUSE_OU] != 0;
        if (!ou_fused || r->opt[MFTX_RAFT_OPT_FUSE_OU] == 2) {         // (2: the fused kernel on the materialised input -- A/B, tests)
            const long long slots = (long long)M * 178;
            ProfScope prof(PC_GLUE, s, 0);
            hipLaunchKernelGGL(ou_gather_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, s, ws.hx,
                               ws.corr, ws.ld_corr, ccur, ws.delta, ws.ouin, flow_lr, M, h, w, SP ? 1 : 0);
            TRY(check_launch("ou_gather"));
        }
        if (ou_fused) {
            // both layers of both heads as ONE tile-resident kernel (five channel passes over the 712-channel input, the 3 x 3 x 3 second
            // layers as a projection epilogue; T -> ws.ouh) + a stencil sum; the input is gathered from its parts by the kernel's loader
            const OuGather ga{ws.hx, ws.corr, ws.ld_corr, ccur, ws.delta, flow_lr};
            const bool mat = r->opt[MFTX_RAFT_OPT_FUSE_OU] == 2;
            TRY(launch_ou_heads(mat ? ws.ouin : nullptr, 712, P, h, w, r->wou, W[B_OU1], r->wouproj, W[B_OU2], ws.ouh, ws.ou, 4, r->opt[MFTX_RAFT_OPT_TILE_CELLS], s,
                                mat ? nullptr : &ga));
        } else {
            TRY(launch_conv(gemm(conv_desc(ws.ouin, 712, 712, nullptr, 0, 0, G[W_OU1], W[B_OU1], ws.ouh, 256, P, h, w, 256, 3, 3, 1), true, false), s));
            TRY(launch_conv(conv_desc(ws.ouh, 256, 256, nullptr, 0, 0, W[W_OU2], W[B_OU2], ws.ou, 4, P, h, w, 3, 3, 3, 0), s));
        }
    }
    return 0;
    };   // core
    if (use_graph) {
        if (AR == MFTX_ARITH_SPLIT && r->opt[MFTX_RAFT_OPT_FORK] != 0) TRY(ensure_side_stream(r));      // (not while capturing)
        GraphKey key{};
        key.v[0] = (uintptr_t)P; key.v[1] = (uintptr_t)h; key.v[2] = (uintptr_t)w; key.v[3] = (uintptr_t)iters;
        key.v[4] = reinterpret_cast<uintptr_t>(workspace); key.v[5] = reinterpret_cast<uintptr_t>(flow_lr_out);
        key.v[6] = (uintptr_t)AR; key.v[7] = reinterpret_cast<uintptr_t>(r->wfused); key.v[8] = reinterpret_cast<uintptr_t>(s);
        key.v[9] = reinterpret_cast<uintptr_t>(r->wflow); key.v[10] = reinterpret_cast<uintptr_t>(r->wt[W_ZR1_DYN]); key.v[11] = reinterpret_cast<uintptr_t>(r->wproj);
        key.v[12] = reinterpret_cast<uintptr_t>(r->wou);
        TRY(r->graphs->run(key, s, core));
    } else {
        TRY(core());
    }
    return launch_convex_upsample(flow_lr, ws.ou, 4, ws.mask, P, h, w, pad_left, pad_right, pad_top, pad_bottom,
                                  flow, occl, sigma, packed, s, r->nonfinite);
}

extern "C" int mftx_raft_refine(mftx_raft *r, int P, int h, int w, int iters, const float *fmap1,
                                const float *fmap2, const float *net, const float *inp, const float *flow_init,
                                int pad_left, int pad_right, int pad_top, int pad_bottom, float *flow, float *occl,
                                float *sigma, float *packed, float *flow_lr_out, void *workspace,
                                size_t workspace_bytes, void *stream) {
    return refine_impl(r, P, h, w, iters, fmap1, fmap2, net, inp, flow_init, pad_left, pad_right, pad_top, pad_bottom, flow, occl, sigma,
                       packed, flow_lr_out, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int mftx_raft_refine_gather(mftx_raft *r, int P, int h, int w, int iters, const float *const *fmap1,
                                       const float *const *fmap2, const float *const *net, const float *const *inp,
                                       const float *flow_init, int pad_left, int pad_right, int pad_top, int pad_bottom,
                                       float *flow, float *occl, float *sigma, float *packed, float *flow_lr_out,
                                       void *workspace, size_t workspace_bytes, void *stream) {
    if (!r || r->magic != RAFT_MAGIC) return fail(MFTX_E_STATE, "raft_refine_gather: bad handle");
    if (!fmap1 || !fmap2 || !net || !inp) return fail(MFTX_E_ARG, "raft_refine_gather: null pointer");
    if (P < 1 || P > MFTX_MAX_GATHER) return fail(MFTX_E_ARG, "raft_refine_gather: 1 .. %d pairs", MFTX_MAX_GATHER);
    if (r->arith != MFTX_ARITH_SPLIT || r->ondemand || r->opt[MFTX_RAFT_OPT_TILE_VOLUME] == 0)
        return fail(MFTX_E_STATE, "raft_refine_gather: needs the split arithmetic with the stored, tile-resident correlation volume (use mftx_raft_refine)");
    RefineGather g{};
    g.f2_shared = true;
    for (int b = 0; b < P; ++b) {
        if (!fmap1[b] || !fmap2[b] || !net[b] || !inp[b]) return fail(MFTX_E_ARG, "raft_refine_gather: pair %d has a null map", b);
        if (!aligned16(fmap1[b]) || !aligned16(fmap2[b]) || !aligned16(net[b]) || !aligned16(inp[b])) return fail(MFTX_E_ALIGN, "raft_refine_gather: maps must be 16-byte aligned");
        g.f1.p[b] = fmap1[b]; g.f2.p[b] = fmap2[b]; g.net.p[b] = net[b]; g.inp.p[b] = inp[b];
        if (fmap2[b] != fmap2[0]) g.f2_shared = false;
    }
    return refine_impl(r, P, h, w, iters, nullptr, nullptr, nullptr, nullptr, flow_init, pad_left, pad_right, pad_top, pad_bottom, flow, occl,
                       sigma, packed, flow_lr_out, workspace, workspace_bytes, stream, &g);
}

extern "C" int mftx_raft_graph_stats(const mftx_raft *r, unsigned long long *captures, unsigned long long *replays) {
    if (!r || r->magic != RAFT_MAGIC || !captures || !replays) return fail(MFTX_E_STATE, "raft_graph_stats: bad arguments");
    *captures = r->graphs ? r->graphs->captures : 0;
    *replays = r->graphs ? r->graphs->replays : 0;
    return 0;
}

// ---------------------------------------------------------------------------
// per-op exports
// ---------------------------------------------------------------------------
extern "C" int mftx_corr_pyramid(const float *f1, const float *f2, int P, int C, int h, int w, float *lvl0,
                                 float *lvl1, float *lvl2, float *lvl3, void *stream) {
    if (!f1 || !f2 || !lvl0 || !lvl1 || !lvl2 || !lvl3) return fail(MFTX_E_ARG, "corr_pyramid: null pointer");
    if (P <= 0 || C <= 0 || C % 32 || h < 8 || w < 8) return fail(MFTX_E_ARG, "corr_pyramid: need C %% 32 == 0, h, w >= 8");
    if (!aligned16(f1) || !aligned16(f2)) return fail(MFTX_E_ALIGN, "corr_pyramid: features must be 16-byte aligned");
    if (!aligned16(lvl0) || !aligned16(lvl1) || !aligned16(lvl2) || !aligned16(lvl3))
        return fail(MFTX_E_ALIGN, "corr_pyramid: levels must be 16-byte aligned");
    float *const lv[4] = {lvl0, lvl1, lvl2, lvl3};
    return launch_corr_pyramid(f1, f2, P, C, h, w, lv, (hipStream_t)stream);
}

extern "C" int mftx_corr_pyramid_split(const float *f1, const float *f2, int P, int C, int h, int w, float *lvl0,
                                       float *lvl1, float *lvl2, float *lvl3, float *f2_scratch, void *stream) {
    if (!f1 || !f2 || !lvl0 || !lvl1 || !lvl2 || !lvl3 || !f2_scratch) return fail(MFTX_E_ARG, "corr_pyramid_split: null pointer");
    if (P <= 0 || C <= 0 || C % 32 || h < 8 || w < 8) return fail(MFTX_E_ARG, "corr_pyramid_split: need C %% 32 == 0, h, w >= 8");
    if (!aligned16(f1) || !aligned16(f2) || (reinterpret_cast<uintptr_t>(f2_scratch) & 31))
        return fail(MFTX_E_ALIGN, "corr_pyramid_split: features must be 16-byte aligned, the scratch 32-byte aligned");
    if (!aligned16(lvl0) || !aligned16(lvl1) || !aligned16(lvl2) || !aligned16(lvl3))
        return fail(MFTX_E_ALIGN, "corr_pyramid_split: levels must be 16-byte aligned");
    float *const lv[4] = {lvl0, lvl1, lvl2, lvl3};
    return launch_corr_pyramid(f1, f2, P, C, h, w, lv, (hipStream_t)stream, f2_scratch);
}

extern "C" int mftx_corr_pyramid_layout(int h, int w, long long *stride, int *block_grid) {
    if (h < 8 || w < 8 || !stride || !block_grid) return fail(MFTX_E_ARG, "corr_pyramid_layout: bad arguments");
    const PyramidLayout L = pyramid_layout(h, w);
    for (int l = 0; l < 4; ++l) stride[l] = L.stride[l];
    block_grid[0] = L.hb[0]; block_grid[1] = L.wb[0]; block_grid[2] = L.hb[1]; block_grid[3] = L.wb[1];
    return 0;
}

extern "C" int mftx_corr_lookup(const float *lvl0, const float *lvl1, const float *lvl2, const float *lvl3,
                                const float *coords, int P, int h, int w, int r, float *out, int ld_out,
                                void *stream) {
    if (!lvl0 || !lvl1 || !lvl2 || !lvl3 || !coords || !out) return fail(MFTX_E_ARG, "corr_lookup: null pointer");
    if (r != 4) return fail(MFTX_E_ARG, "corr_lookup: only radius 4 (RAFT basic) is built");
    if (P <= 0 || h < 8 || w < 8 || ld_out < 324) return fail(MFTX_E_ARG, "corr_lookup: bad sizes");
    if (!aligned16(lvl0) || !aligned16(lvl1)) return fail(MFTX_E_ALIGN, "corr_lookup: levels 0 and 1 must be 16-byte aligned");
    const float *lv[4] = {lvl0, lvl1, lvl2, lvl3};
    return launch_corr_lookup(lv, coords, P, h, w, out, ld_out, (hipStream_t)stream);
}

extern "C" int mftx_pack_lookup_convc1_weights(const float *wpk, int ld_w, void *wfused, void *stream) {
    if (!wpk || !wfused) return fail(MFTX_E_ARG, "pack_lookup_convc1_weights: null pointer");
    if (ld_w < 324) return fail(MFTX_E_ARG, "pack_lookup_convc1_weights: need ld_w >= 324");
    if (!aligned16(wfused)) return fail(MFTX_E_ALIGN, "pack_lookup_convc1_weights: output must be 16-byte aligned");
    return launch_pack_lookup_convc1(wpk, ld_w, wfused, (hipStream_t)stream);
}

extern "C" int mftx_corr_lookup_convc1(const float *lvl0, const float *lvl1, const float *lvl2, const float *lvl3,
                                       const float *coords, int P, int h, int w, const void *wfused, const float *bias,
                                       float *out, int ld_out, int out_split, void *stream) {
    if (!lvl0 || !lvl1 || !lvl2 || !lvl3 || !coords || !wfused || !bias || !out) return fail(MFTX_E_ARG, "corr_lookup_convc1: null pointer");
    if (P <= 0 || h < 8 || w < 8 || ld_out < 256 || ld_out % 4) return fail(MFTX_E_ARG, "corr_lookup_convc1: bad sizes");
    if (!lookup_convc1_applicable(P, h, w, ld_out)) return fail(MFTX_E_ARG, "corr_lookup_convc1: batch too large");
    if (!aligned16(wfused) || !aligned16(bias) || !aligned16(out) || !aligned16(coords) ||
        (out_split && (ld_out % 8 || (reinterpret_cast<uintptr_t>(out) & 31))))
        return fail(MFTX_E_ALIGN, "corr_lookup_convc1: operands must be 16-byte aligned (a split-form output: 32-byte rows)");
    const float *lv[4] = {lvl0, lvl1, lvl2, lvl3};
    return launch_lookup_convc1(lv, coords, P, h, w, wfused, bias, out, ld_out, out_split ? 1 : 0, (hipStream_t)stream);
}

extern "C" int mftx_pack_tile_conv_weights(const float *wpk, int N, int taps, int cin, int cin_pad, void *wtile, void *stream) {
    if (taps == 9 && cin == 256) return launch_pack_tile_conv2p(wpk, N, cin_pad, wtile, (hipStream_t)stream);      // 3 x 3 over 256 channels: two channel passes
    return launch_pack_tile_conv(wpk, N, taps, cin, cin_pad, wtile, (hipStream_t)stream);
}

extern "C" int mftx_tile_conv2d(const mftx_conv_desc *d, const void *wtile, void *stream) {
    if (!d || !wtile) return fail(MFTX_E_ARG, "tile_conv2d: null pointer");
    if (d->arith != MFTX_ARITH_SPLIT || !d->a_split) return fail(MFTX_E_ARG, "tile_conv2d: split arithmetic with a split-form input only");
    if (d->act != 0 && d->act != 1) return fail(MFTX_E_ARG, "tile_conv2d: activation none or relu");
    if (d->stride > 1 || d->residual_mode != 0 || d->out_scale != 1.f || (d->hin && d->hin != d->h) || (d->win && d->win != d->w) || d->pad_y != 0 || d->pad_x != 0)
        return fail(MFTX_E_ARG, "tile_conv2d: stride 1, same padding, no output scale");
    if (d->kh == 3 && d->kw == 3 && d->c0 == 256 && d->c1 == 0) {       // two channel passes (tile_conv.hip: tile_conv2p_kernel)
        if (d->act != 1 || !d->out_split || d->addend || !d->bias) return fail(MFTX_E_ARG, "tile_conv2d: 3 x 3 over 256 channels comes with bias, relu and a split-form output");
        return launch_tile_conv2p(d->a0, d->lda0, wtile, d->bias, d->out, d->ldo, d->N, d->P, d->h, d->w, 0, (hipStream_t)stream);
    }
    if (d->c0 != 128 || (d->c1 != 0 && d->c1 != 128)) return fail(MFTX_E_ARG, "tile_conv2d: channel segments of 128");
    TileConvLaunch t{};
    t.a0 = d->a0; t.lda0 = d->lda0; t.a1 = d->c1 ? d->a1 : nullptr; t.lda1 = d->lda1; t.cin = d->c0 + d->c1; t.wf = wtile; t.bias = d->bias;
    t.addend = d->addend; t.ld_addend = d->ld_addend; t.out = d->out; t.ldo = d->ldo; t.out_split = d->out_split;
    t.P = d->P; t.h = d->h; t.w = d->w; t.N = d->N; t.kh = d->kh; t.kw = d->kw; t.epi = d->act;
    return launch_tile_conv(t, (hipStream_t)stream);
}

extern "C" int mftx_gru_half(const float *h_in, int ld_hin, const float *motion, int ld_mo, const void *wzr, const void *wq, const float *pre_zr,
                             const float *pre_q, float *z, const float *hf_in, float *hf_out, float *h_out, int ld_hout, int P, int h, int w, int pass,
                             void *stream) {
    GruHalfLaunch g{};
    g.h_in = h_in; g.ld_hin = ld_hin; g.mo = motion; g.ld_mo = ld_mo; g.wzr = wzr; g.wq = wq; g.pre_zr = pre_zr; g.pre_q = pre_q;
    g.z = z; g.hf_in = hf_in; g.hf_out = hf_out; g.h_out = h_out; g.ld_hout = ld_hout; g.P = P; g.h = h; g.w = w; g.pass = pass;
    return launch_gru_half(g, (hipStream_t)stream);
}

extern "C" int mftx_tile_conv_fills_chip(int P, int h, int w) {
    if (P <= 0 || h <= 0 || w <= 0) return 0;
    // (round 4) the tile-resident kernels come with 128, 64 or 32 cells per tile -- the same bits --, so "fills the chip" is asked of the
    // smallest: at least half a round of 32-cell tiles (7 pairs of 256 x 256 pixels: 224 tiles, + 9 % frames/s over the ring-buffered
    // kernels; one pair of 512 x 512: 128 tiles, 2.38 vs 2.46 ms per refinement)
    return tile_conv_fills_chip(P, h, w, 3, 3) || tile_conv_small_tiles_fill(P, h, w) ? 1 : 0;
}

extern "C" int mftx_pack_ou_heads_weights(const float *w1pk, int cin_pad, const float *w2pk, void *wtile, void *wproj, void *stream) {
    return launch_pack_ou_head(w1pk, cin_pad, w2pk, wtile, wproj, (hipStream_t)stream);
}

extern "C" int mftx_ou_heads(const float *a_split, int lda, int P, int h, int w, const void *wtile, const float *b1, const void *wproj, const float *b2,
                             float *T, float *out, int ld_out, void *stream) {
    return launch_ou_heads(a_split, lda, P, h, w, wtile, b1, wproj, b2, T, out, ld_out, 0, (hipStream_t)stream);
}

extern "C" int mftx_pack_flow_head_weights(const float *w2pk, void *wproj, void *stream) {
    return launch_pack_flow_head(w2pk, wproj, (hipStream_t)stream);
}

extern "C" int mftx_flow_head(const float *hsplit, int ld_h, int P, int h, int w, const void *wtile, const float *b1, const void *wproj,
                              const float *b2, float *T, float *delta, float *coords, void *stream) {
    if (!hsplit || !wtile || !b1 || !wproj || !b2 || !T || !delta) return fail(MFTX_E_ARG, "flow_head: null pointer");
    TileConvLaunch t{};
    t.a0 = hsplit; t.lda0 = ld_h; t.cin = 128; t.wf = wtile; t.bias = b1; t.wproj = wproj; t.tout = T;
    t.P = P; t.h = h; t.w = w; t.N = 256; t.kh = 3; t.kw = 3; t.epi = 4;
    if (int e = launch_tile_conv(t, (hipStream_t)stream)) return e;
    return launch_flow_head_sum(T, b2, delta, coords, coords, P, h, w, (hipStream_t)stream);
}

extern "C" int mftx_pack_flow_branch_weights(const float *w98, const float *w2pk, void *wflow, void *stream) {
    if (!w98 || !w2pk || !wflow) return fail(MFTX_E_ARG, "pack_flow_branch_weights: null pointer");
    if (!aligned16(wflow)) return fail(MFTX_E_ALIGN, "pack_flow_branch_weights: output not 16-byte aligned");
    return launch_pack_flow_branch(w98, w2pk, wflow, (hipStream_t)stream);
}

extern "C" int mftx_flow_branch(const float *coords, int P, int h, int w, const void *wflow, const float *b1, const float *b2,
                                float *out, int ld_out, float *hx, int ld_hx, void *stream) {
    return launch_flow_branch(coords, P, h, w, wflow, b1, b2, out, ld_out, hx, ld_hx, (hipStream_t)stream);
}

extern "C" int mftx_fmap_pyramid(const float *f2, int P, int C, int h, int w, float *lvl1, float *lvl2, float *lvl3,
                                 void *stream) {
    if (!f2 || !lvl1 || !lvl2 || !lvl3) return fail(MFTX_E_ARG, "fmap_pyramid: null pointer");
    if (P <= 0 || C <= 0 || C % 4 || h < 8 || w < 8) return fail(MFTX_E_ARG, "fmap_pyramid: need C %% 4 == 0, h, w >= 8");
    if (!aligned16(f2) || !aligned16(lvl1) || !aligned16(lvl2) || !aligned16(lvl3))
        return fail(MFTX_E_ALIGN, "fmap_pyramid: maps must be 16-byte aligned");
    float *const lv[3] = {lvl1, lvl2, lvl3};
    return launch_fmap_pyramid(f2, P, C, h, w, lv, (hipStream_t)stream);
}

extern "C" int mftx_corr_lookup_ondemand(const float *f1, const float *f2l0, const float *f2l1, const float *f2l2,
                                         const float *f2l3, const float *coords, int P, int C, int h, int w, int r,
                                         float *out, int ld_out, void *stream) {
    if (!f1 || !f2l0 || !f2l1 || !f2l2 || !f2l3 || !coords || !out) return fail(MFTX_E_ARG, "corr_lookup_ondemand: null pointer");
    if (r != 4 || C != 256) return fail(MFTX_E_ARG, "corr_lookup_ondemand: only radius 4, 256 channels (RAFT basic) are built");
    if (P <= 0 || h < 8 || w < 8 || ld_out < 324) return fail(MFTX_E_ARG, "corr_lookup_ondemand: bad sizes");
    if (!aligned16(f1) || !aligned16(f2l0) || !aligned16(f2l1) || !aligned16(f2l2) || !aligned16(f2l3))
        return fail(MFTX_E_ALIGN, "corr_lookup_ondemand: feature maps must be 16-byte aligned");
    const float *lv[4] = {f2l0, f2l1, f2l2, f2l3};
    return launch_corr_ondemand(f1, lv, coords, P, h, w, out, ld_out, (hipStream_t)stream);
}

extern "C" int mftx_split_weights(const float *wpk, void *out, long long n_floats, void *stream) {
    if (!wpk || !out) return fail(MFTX_E_ARG, "split_weights: null pointer");
    return launch_split_weights(wpk, out, n_floats, (hipStream_t)stream);
}

extern "C" int mftx_conv2d(const mftx_conv_desc *d, void *stream) {
    if (!d) return fail(MFTX_E_ARG, "conv2d: null descriptor");
    return launch_conv(*d, (hipStream_t)stream);
}

extern "C" int mftx_conv2d_tile(const mftx_conv_desc *d, int tile, void *stream) {
    if (!d) return fail(MFTX_E_ARG, "conv2d_tile: null descriptor");
    if (tile < -1 || tile > 15) return fail(MFTX_E_ARG, "conv2d_tile: tile must be -1 (the library's choice) or 0..15");
    return launch_conv(*d, (hipStream_t)stream, tile);
}

extern "C" int mftx_convex_upsample(const float *flow_lr, const float *ou, int ld_ou, const float *mask, int P,
                                    int h, int w, int pad_left, int pad_right, int pad_top, int pad_bottom,
                                    float *flow, float *occl, float *sigma, float *packed, void *stream) {
    const bool planar = flow && occl && sigma;
    if (!flow_lr || !ou || !mask || (!planar && (flow || occl || sigma || !packed)))
        return fail(MFTX_E_ARG, "convex_upsample: null pointer (outputs: flow + occl + sigma, or packed, or both)");
    if (P <= 0 || h <= 0 || w <= 0 || ld_ou < 3) return fail(MFTX_E_ARG, "convex_upsample: bad sizes");
    return launch_convex_upsample(flow_lr, ou, ld_ou, mask, P, h, w, pad_left, pad_right, pad_top, pad_bottom, flow,
                                  occl, sigma, packed, (hipStream_t)stream);
}
