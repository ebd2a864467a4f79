"""Lane-level numpy model of v_mfma_f32_16x16x4_f32 and of the kernels' chained-GEMM register
layout (uforecon_amd/csrc/ufr_layout.h, ufr_device.h).  Lets the CPU test-suite check the packed
weight plan exported by libufr.so (ufr_pack_plan) without a GPU.

MFMA semantics (cdna_hip_programming.md section 3): D[i][j] += sum_k A[i][k] B[k][j] with lane l
supplying A[l&15][l>>4], B[l>>4][l&15] and owning D[4*(l>>4)+r][l&15], r = 0..3.
"""
from __future__ import annotations

import numpy as np

LANE = np.arange(64)
G, J = LANE >> 4, LANE & 15

# mirrors of the enums in ufr_layout.h
ROW_NAT, ROW_SLOT20, ROW_HEAD11K, ROW_NAT88, ROW_QUAD11 = range(5)
COL_NAT, COL_SLOT20, COL_NAT88, COL_QUAD11, COL_RW0, COL_CAT88 = range(6)

# (name, param, k_raw, n_out, n_in, rm, cm, out_dim, in_dim) in blob order
MATS = [
    ("VT_Q", 6, 80, 5, 5, ROW_SLOT20, COL_NAT, 80, 80), ("VT_K", 7, 80, 5, 5, ROW_SLOT20, COL_NAT, 80, 80),
    ("VT_V", 8, 80, 5, 5, ROW_SLOT20, COL_NAT, 80, 80), ("VT_MERGE", 9, 80, 5, 5, ROW_NAT, COL_SLOT20, 80, 80),
    ("VT_MLP0", 10, 160, 10, 10, ROW_NAT, COL_NAT, 160, 160), ("VT_MLP2", 11, 160, 5, 10, ROW_NAT, COL_NAT, 80, 160),
    ("RT_Q", 16, 88, 6, 6, ROW_QUAD11, COL_NAT88, 88, 88), ("RT_K", 17, 88, 8, 6, ROW_HEAD11K, COL_NAT88, 88, 88),
    ("RT_V", 18, 88, 8, 6, ROW_HEAD11K, COL_NAT88, 88, 88), ("RT_MERGE", 19, 88, 6, 6, ROW_NAT88, COL_QUAD11, 88, 88),
    ("RT_MLP0", 20, 176, 11, 12, ROW_NAT, COL_CAT88, 176, 176), ("RT_MLP2", 21, 176, 6, 11, ROW_NAT88, COL_NAT, 88, 176),
    ("DM0", 26, 88, 2, 6, ROW_NAT, COL_NAT88, 32, 88), ("DM2", 28, 32, 1, 2, ROW_NAT, COL_NAT, 16, 32),
    ("DM4", 30, 16, 1, 1, ROW_NAT, COL_NAT, 1, 16), ("RW0", 32, 83, 1, 6, ROW_NAT, COL_RW0, 16, 83),
    ("RW2", 34, 16, 1, 1, ROW_NAT, COL_NAT, 8, 16), ("RW4", 36, 8, 1, 1, ROW_NAT, COL_NAT, 1, 8),
]


def nat88(t, g, r):
    return 16 * t + 4 * g + r if t < 5 else (80 + 2 * g + r if r < 2 else -1)


def head11_slot(i):
    """head dim held by slot i = 4g + r of a 16-slot head tile (ufr_layout.h): 3g + r for r < 3, padding otherwise."""
    g, r = i >> 2, i & 3
    return 3 * g + r if (r < 3 and 3 * g + r < 11) else -1


def quad11(t, g, r):
    """feature of register r, lane group g of quad-packed tile t: quad 4t + r = (head, live register)."""
    h, q = divmod(4 * t + r, 3)
    return 11 * h + 3 * g + q if 3 * g + q < 11 else -1


def row_map(rm, t, i, out_dim):
    g, r = i >> 2, i & 3
    v = {ROW_NAT: 16 * t + i, ROW_SLOT20: 20 * g + 4 * t + r,
         ROW_HEAD11K: 11 * t + head11_slot(i) if head11_slot(i) >= 0 else -1,
         ROW_NAT88: nat88(t, g, r), ROW_QUAD11: quad11(t, g, r)}[rm]
    return v if 0 <= v < out_dim else -1


def col_map(cm, t, g, r, in_dim):
    if cm == COL_NAT:
        v = 16 * t + 4 * g + r
    elif cm == COL_SLOT20:
        v = 20 * g + 4 * t + r
    elif cm == COL_NAT88:
        v = nat88(t, g, r)
    elif cm == COL_QUAD11:
        v = quad11(t, g, r)
    elif cm == COL_RW0:
        v = 16 * t + 4 * g + r if t < 5 else (80 + g if (r == 0 and g < 3) else -1)
    else:
        v = nat88(t, g, r) if t < 6 else (-1 if nat88(t - 6, g, r) < 0 else 88 + nat88(t - 6, g, r))
    return v if 0 <= v < in_dim else -1


def in_steps(cm, t):
    if cm == COL_NAT88 and t == 5:
        return 2
    if cm == COL_CAT88 and t in (5, 11):
        return 2
    if cm == COL_RW0 and t == 5:
        return 1
    return 4


def mfma16(a, b, acc):
    """a, b: (64,) lane operands; acc: (64,4) accumulator registers (updated copy returned)."""
    A = np.zeros((16, 4), np.float64)
    B = np.zeros((4, 16), np.float64)
    A[J, G] = a
    B[G, J] = b
    D = A @ B
    out = acc.copy()
    for r in range(4):
        out[:, r] += D[4 * G + r, J]
    return out


# weight streams (ufr_layout.h): (matrix names in consumption order, out tiles interleaved per stage)
STREAMS = [
    (["VT_Q", "VT_K", "VT_V", "VT_MERGE", "VT_MLP0", "VT_MLP2", "RW0", "RW2", "RW4"], 1),
    (["RT_K", "RT_V"], 2),
    (["RT_Q", "RT_MERGE", "RT_MLP0", "RT_MLP2", "DM0", "DM2", "DM4"], 2),
]
NAME2IDX = {m[0]: i for i, m in enumerate(MATS)}


def stream_frags_padded(si):
    n = sum(MATS[NAME2IDX[m]][3] * MATS[NAME2IDX[m]][4] for m in STREAMS[si][0])
    return (n + 7) // 8 * 8  # kFetchSplit of ufr_layout.h


def vec_region_offset():
    return sum(stream_frags_padded(i) for i in range(len(STREAMS))) * 256


def mat_location(idx):
    """(blob float offset of the matrix, OT of its stream)."""
    name = MATS[idx][0]
    base = 0
    for si, (names, ot) in enumerate(STREAMS):
        if name in names:
            off = sum(MATS[NAME2IDX[m]][3] * MATS[NAME2IDX[m]][4] for m in names[:names.index(name)])
            return base + off * 256, ot
        base += stream_frags_padded(si) * 256
    raise KeyError(name)


def frag_in_mat(idx, ot, to, ti):
    n_out, n_in = MATS[idx][3], MATS[idx][4]
    gi, o = divmod(to, ot)
    no = min(ot, n_out - gi * ot)
    return gi * ot * n_in + ti * no + o


def gemm(blob, idx, tiles_in, swap=False):
    """Chained GEMM of matrix `idx`: tiles_in (n_in,64,4) registers -> (n_out,64,4) accumulators."""
    _, _, _, n_out, n_in, _, cm, _, _ = MATS[idx]
    base, ot = mat_location(idx)
    out = np.zeros((n_out, 64, 4), np.float64)
    for to in range(n_out):
        for ti in range(n_in):
            f = base + frag_in_mat(idx, ot, to, ti) * 256
            frag = blob[f: f + 256].reshape(64, 4)
            for r in range(in_steps(cm, ti)):
                a, b = frag[:, r], tiles_in[ti][:, r]
                out[to] = mfma16(b, a, out[to]) if swap else mfma16(a, b, out[to])
    return out


def to_tiles(x, cm, n_tiles, in_dim):
    """x (16 tokens, in_dim) -> B-operand registers (n_tiles,64,4): lane (g,j) reg r <- x[j][col(t,g,r)]."""
    t = np.zeros((n_tiles, 64, 4), np.float64)
    for ti in range(n_tiles):
        for lane in range(64):
            for r in range(4):
                c = col_map(cm, ti, lane >> 4, r, in_dim)
                if c >= 0:
                    t[ti, lane, r] = x[lane & 15, c]
    return t


def from_tiles(acc, rm, out_dim):
    """accumulators (n_out,64,4) in standard orientation -> y (16 tokens, out_dim); padding rows must be 0."""
    y = np.zeros((16, out_dim), np.float64)
    pad_abs = 0.0
    for to in range(acc.shape[0]):
        for lane in range(64):
            for r in range(4):
                row = row_map(rm, to, 4 * (lane >> 4) + r, out_dim)
                if row >= 0:
                    y[lane & 15, row] = acc[to, lane, r]
                else:
                    pad_abs = max(pad_abs, abs(acc[to, lane, r]))
    return y, pad_abs


def from_tiles_swapped(acc, rm, out_dim):
    """swapped orientation: lane (g,j) reg r = y[token 4g+r][row(to, j)]."""
    y = np.zeros((16, out_dim), np.float64)
    for to in range(acc.shape[0]):
        for lane in range(64):
            row = row_map(rm, to, lane & 15, out_dim)
            if row >= 0:
                for r in range(4):
                    y[4 * (lane >> 4) + r, row] = acc[to, lane, r]
    return y


# ------------------------------------------------------------------ fp16x3 path (ufr_layout_f16.h, weight_stream_f16.h)
W_SCALE, X_SCALE = np.float32(256.0), np.float32(16.0)   # kWScale, kXScale


def split2(x, scale):
    """fp16 planes of scale * x (ufr_layout_f16.h): hi = fp16(scale x), lo = fp16(scale x - hi), RNE; returned as float32."""
    x = np.asarray(x, np.float32) * np.float32(scale)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def f16_streams():
    """Per weight stream: (matrix name, k-step) panels in stream order (ufr_layout_f16.h:f16_panel)."""
    vt = []
    for s in range(3):
        vt += [("VT_Q", s), ("VT_K", s)]
    vt += [("VT_V", s) for s in range(3)] + [("VT_MERGE", s) for s in range(3)]
    vt += [("VT_MLP0", s) for s in range(5)] + [("VT_MLP2", s) for s in range(5)]
    vt += [("RW0", s) for s in range(3)] + [("RW2", 0), ("RW4", 0)]
    rt1 = []
    for s in range(3):
        rt1 += [("RT_K", s), ("RT_V", s)]
    rt2 = [("RT_Q", s) for s in range(3)] + [("RT_MERGE", s) for s in range(3)]
    rt2 += [("RT_MLP0", s) for s in range(6)] + [("RT_MLP2", s) for s in range(6)]
    rt2 += [("DM0", s) for s in range(3)] + [("DM2", 0), ("DM4", 0)]
    return [vt, rt1, rt2]


F16_CHUNK = 12  # fragments per LDS chunk (kF16ChunkFrags)
F16_SLOTS = 3   # LDS ring depth (kF16Slots): streams are padded to whole rings


def _stream_len(panels):
    n = sum(MATS[NAME2IDX[m]][3] * 2 for m, _ in panels)
    chunks = (n + F16_CHUNK - 1) // F16_CHUNK
    return (chunks + F16_SLOTS - 1) // F16_SLOTS * F16_SLOTS * F16_CHUNK


def panel_start(name, s):
    """first fragment of panel (name, k-step s) in the fp16 plane region: streams are padded to whole chunks."""
    base = 0
    for panels in f16_streams():
        off = 0
        for n, k in panels:
            if (n, k) == (name, s):
                return base + off
            off += MATS[NAME2IDX[n]][3] * 2
        base += _stream_len(panels)
    raise KeyError((name, s))


def f16_region_frags():
    return sum(_stream_len(p) for p in f16_streams())


def mfma_f16(a, b, acc):
    """v_mfma_f32_16x16x32_f16: a, b (64,8) lane operands (lane l: A[l&15][8(l>>4)+i], B[8(l>>4)+i][l&15])."""
    A = np.zeros((16, 32), np.float64)
    B = np.zeros((32, 16), np.float64)
    for i in range(8):
        A[J, 8 * G + i] = a[:, i]
        B[8 * G + i, J] = b[:, i]
    D = A @ B
    out = acc.astype(np.float64).copy()
    for r in range(4):
        out[:, r] += D[4 * G + r, J]
    return out.astype(np.float32)  # the accumulator is fp32


def gemm_f16(f16_blob, name, tiles_in, swap=False):
    """tiles_in (n_in,64,4) fp32 accumulator tiles of the producer -> (n_out,64,4); swap: activations in the A slot."""
    idx = NAME2IDX[name]
    n_out, n_in = MATS[idx][3], MATS[idx][4]
    out = np.zeros((n_out, 64, 4), np.float32)
    zero = np.zeros((64, 4), np.float32)
    for s in range((n_in + 1) // 2):
        ta, tb = tiles_in[2 * s], (tiles_in[2 * s + 1] if 2 * s + 1 < n_in else zero)
        xb = [np.concatenate([pa, pb], axis=1) for pa, pb in zip(split2(ta, X_SCALE), split2(tb, X_SCALE))]  # per plane (64,8)
        f0 = panel_start(name, s)
        for to in range(n_out):
            a = [f16_blob[(f0 + to * 2 + p) * 512:(f0 + to * 2 + p + 1) * 512].reshape(64, 8) for p in range(2)]
            for pa, pb in ((1, 0), (0, 1), (0, 0)):     # w_lo x_hi, w_hi x_lo, w_hi x_hi
                out[to] = mfma_f16(xb[pb], a[pa], out[to]) if swap else mfma_f16(a[pa], xb[pb], out[to])
    return out * np.float32(1.0 / 4096.0)               # kAccDescale
