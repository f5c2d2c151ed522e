"""ctypes front end of oracle/libmft_oracle_c.so (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
from pathlib import Path

import numpy as np

_PATH = Path(__file__).resolve().parent / "libmft_oracle_c.so"
_lib = None
_F = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(str(_PATH))
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def corr_volume(f1, f2):
    """f1, f2: [C, N] -> [N, N]"""
    Cc, N = f1.shape
    out = np.empty((N, N), np.float32)
    load().orc_corr_volume(_fp(f1), _fp(f2), Cc, N, _fp(out))
    return out


def avg_pool2(src, rows, h, w):
    out = np.empty((rows, h // 2, w // 2), np.float32)
    load().orc_avg_pool2(_fp(src), rows, h, w, _fp(out))
    return out


def corr_lookup(levels, sizes, coords, r=4):
    """levels: list of [N, hl*wl]; sizes [(hl, wl)]; coords [2, N] -> [L*81, N]"""
    L, N = len(levels), coords.shape[1]
    ptrs = (C.POINTER(C.c_float) * L)(*[_fp(l) for l in levels])
    hl = (C.c_int * L)(*[s[0] for s in sizes])
    wl = (C.c_int * L)(*[s[1] for s in sizes])
    out = np.empty((L * (2 * r + 1) ** 2, N), np.float32)
    load().orc_corr_lookup(ptrs, hl, wl, L, r, _fp(coords), N, _fp(out))
    return out


def conv2d(x, w, b):
    Cin, H, W = x.shape
    Cout, _, kh, kw = w.shape
    out = np.empty((Cout, H, W), np.float32)
    load().orc_conv2d(_fp(x), Cin, H, W, _fp(w), _fp(b) if b is not None else None, Cout, kh, kw, _fp(out))
    return out


def convex_upsample(x, mask, mult):
    Cc, h, w = x.shape
    out = np.empty((Cc, 8 * h, 8 * w), np.float32)
    load().orc_convex_upsample(_fp(x), Cc, h, w, _fp(mask), C.c_float(mult), _fp(out))
    return out


def chain(L, R):
    _, H, W = L[0].shape
    out = (np.empty((2, H, W), np.float32), np.empty((1, H, W), np.float32), np.empty((1, H, W), np.float32))
    load().orc_chain(*[_fp(a) for a in L], *[_fp(a) for a in R], H, W, *[_fp(a) for a in out])
    return out


def select(cands, thr):
    K = len(cands)
    _, H, W = cands[0][0].shape
    cols = [(C.POINTER(C.c_float) * K)(*[_fp(c[j]) for c in cands]) for j in range(3)]
    out = (np.empty((2, H, W), np.float32), np.empty((1, H, W), np.float32), np.empty((1, H, W), np.float32))
    chosen = np.empty((H, W), np.int8)
    load().orc_select(K, cols[0], cols[1], cols[2], C.c_float(thr), H, W, *[_fp(a) for a in out],
                      chosen.ctypes.data_as(C.POINTER(C.c_int8)))
    return out + (chosen,)
