"""CPU-only checks of the C ABI surface and of the packed-weight layout (no compute on a GPU).

* libufr.so loads and exports every symbol include/ufr.h declares;
* argument validation returns error codes + messages instead of crashing;
* the weight re-ordering plan exported by the library (ufr_pack_plan), pushed through a lane-level
  model of v_mfma_f32_16x16x4_f32 (tests/mfma_emu.py), reproduces y = W x for every matrix -- i.e.
  the "accumulator tile == next B operand" chaining and all row/column permutations are right.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

import mfma_emu as E
from uforecon_amd import _lib, ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from uforecon_amd.build import build_library

        build_library(verbose=False)
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "ufr.h")).read()
    declared = set(re.findall(r"\b(ufr_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ufr_version() == _lib.ABI_VERSION


def test_struct_sizes_match_the_header(lib):
    assert C.sizeof(_lib.RawWeights) == 40 * 8
    assert C.sizeof(_lib.Frame) == 160 * 8
    n_vec = 16 * (5 * 4 + 6 * 4 + 2 + 5 + 5 + 1 + 1)   # + rw4 and dm4 as vectors
    assert lib.ufr_packed_scale_table_offset() == E.vec_region_offset() + n_vec
    assert lib.ufr_packed_scale_table_entries() == 18          # one {2^a, 2^-(s+a), 2^(s+a), 2^s} per dense matrix
    # + the two kernels' scalar lists + the weight statistics the exponents derive from (kept for ufr_weights_fit_frame)
    assert lib.ufr_packed_fp32_floats() == E.vec_region_offset() + n_vec + 4 * 18 + 2 * 32 + 3 * 32
    assert lib.ufr_packed_weights_bytes() == (4 * lib.ufr_packed_fp32_floats() + 2 * lib.ufr_packed_f16_halfwords()
                                              + 2 * lib.ufr_packed_bwd_halfwords() + 16)  # + flag tail
    assert lib.ufr_packed_bwd_halfwords() % (12 * 512) == 0
    assert lib.ufr_packed_f16_halfwords() % (12 * 512) == 0  # whole 12 KiB chunks


def test_argument_errors_are_reported_not_fatal(lib):
    assert lib.ufr_sample_fixed(None, None, None, None, 4, 64, None) == -1
    assert b"null" in lib.ufr_last_error()
    d = _lib.FrameDesc()
    d.NV = 9
    assert lib.ufr_frame_workspace_bytes(C.byref(d)) == 0
    assert b"NV=9" in lib.ufr_last_error()
    assert lib.ufr_aggregate(None, None, None, None, 4, 60, 3, None, None, None, None, None, -1, None) == -1
    fr = _lib.Frame()  # never prepared -> magic missing
    assert lib.ufr_project_gather(C.byref(fr), None, None, 0, None, None, 1, 16, None, None, None, None, None, None,
                                  None, None, None, None) == -1
    assert b"not prepared" in lib.ufr_last_error()
    assert lib.ufr_composite_bwd(None, None, None, None, None, 4, 64, None, None, None, None, None, 0, None, None, None) == -1
    assert lib.ufr_render_loss(None, None, None, None, None, None, None, 2, 1, 4, 1.0, 1.0, None, None, None, None, None, None) == -1
    assert lib.ufr_aggregate_bwd(None, None, None, None, None, None, None, 4, 64, 3, None, None, None, None, -1, None) == -1
    assert b"null" in lib.ufr_last_error()
    assert lib.ufr_project_gather_bwd(C.byref(fr), None, None, None, 0, None, None, 1, 16, None, None, None, None, None, 0, None, -1, None) == -1
    assert b"not prepared" in lib.ufr_last_error()


def test_missing_gpu_is_loud():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ops.UfrError, match="no CPU implementation"):
        ops.sample_fixed(torch.zeros(4), torch.ones(4), torch.rand(64, 4))


@pytest.fixture(scope="module")
def plan(lib):
    n = lib.ufr_packed_fp32_floats()
    pid = np.zeros(n, np.int32)
    el = np.zeros(n, np.int32)
    assert lib.ufr_pack_plan(pid.ctypes.data_as(C.POINTER(C.c_int32)), el.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    return pid, el


def _blob(plan, raw):
    pid, el = plan
    blob = np.zeros(pid.shape[0], np.float64)
    for p in np.unique(pid):
        if p < 0:
            continue
        sel = pid == p
        blob[sel] = raw[p].reshape(-1)[el[sel]]
    return blob


@pytest.fixture(scope="module")
def raw_and_blob(plan):
    rng = np.random.default_rng(0)
    raw = {i: rng.standard_normal(int(np.prod(s)) if s else 1) for i, s in enumerate(ops.RAW_WEIGHT_SHAPES)}
    return raw, _blob(plan, raw)


@pytest.mark.parametrize("idx", range(len(E.MATS)))
def test_chained_mfma_gemm_reproduces_linear(idx, raw_and_blob):
    raw, blob = raw_and_blob
    name, param, k_raw, n_out, n_in, rm, cm, out_dim, in_dim = E.MATS[idx]
    W = raw[param].reshape(out_dim, k_raw)
    rng = np.random.default_rng(100 + idx)
    x = rng.standard_normal((16, in_dim))
    acc = E.gemm(blob, idx, E.to_tiles(x, cm, n_in, in_dim))
    y, pad = E.from_tiles(acc, rm, out_dim)
    np.testing.assert_allclose(y, x @ W[:, :in_dim].T, rtol=1e-12, atol=1e-12, err_msg=name)
    assert pad == 0.0
    if name in ("RT_K", "RT_V"):  # operands swapped: result tile is [token][head dim]
        acc = E.gemm(blob, idx, E.to_tiles(x, cm, n_in, in_dim), swap=True)
        np.testing.assert_allclose(E.from_tiles_swapped(acc, rm, out_dim), x @ W.T, rtol=1e-12, atol=1e-12)


def test_vector_fragments(plan, raw_and_blob):
    raw, blob = raw_and_blob
    off = E.vec_region_offset()
    vecs = [(12, 5, E.ROW_NAT, 80), (13, 5, E.ROW_NAT, 80), (14, 5, E.ROW_NAT, 80), (15, 5, E.ROW_NAT, 80),
            (22, 6, E.ROW_NAT88, 88), (23, 6, E.ROW_NAT88, 88), (24, 6, E.ROW_NAT88, 88), (25, 6, E.ROW_NAT88, 88),
            (27, 2, E.ROW_NAT, 32), (29, 1, E.ROW_NAT, 16), (31, 1, E.ROW_NAT, 1), (33, 1, E.ROW_NAT, 16),
            (35, 1, E.ROW_NAT, 8), (37, 1, E.ROW_NAT, 1), (38, 5, E.ROW_NAT, 80), (36, 1, E.ROW_NAT, 8), (30, 1, E.ROW_NAT, 16)]
    for param, nt, rm, dim in vecs:
        frag = blob[off: off + nt * 16].reshape(nt, 4, 4)
        for t in range(nt):
            for g in range(4):
                for r in range(4):
                    row = E.row_map(rm, t, 4 * g + r, dim)
                    assert frag[t, g, r] == (raw[param][row] if row >= 0 else 0.0)
        off += nt * 16
    assert off + 4 * 18 + 2 * 32 + 3 * 32 == blob.shape[0]   # then the scale table, scalar lists, statistics: no parameter maps there (the pack kernel derives them)
    assert not blob[off:].any()


def test_ray_attention_dataflow_in_mfma_form(raw_and_blob):
    """KV = K'^T V, message = (Q' KV) / (Q'.sum K') and the merge projection computed exactly as ray_transformer.hip does
    (swapped K / V projections with the 11 head dims in the slots 4g + r, r < 3; ones column in the padding slot 3;
    quad-packed Q: three MFMAs per head; the message's live registers renamed into merge's 6 input tiles) equal the
    textbook linear attention followed by the merge matrix."""
    raw, blob = raw_and_blob
    rng = np.random.default_rng(7)
    SN = 32
    X = rng.standard_normal((SN, 88))
    Wq, Wk, Wv, Wm = (raw[p].reshape(88, 88) for p in (16, 17, 18, 19))
    elu1 = lambda a: np.where(a > 0, a + 1, np.exp(a))
    iq, ik, iv, im = (E.NAME2IDX[n] for n in ("RT_Q", "RT_K", "RT_V", "RT_MERGE"))
    slot_ok = np.array([E.head11_slot(int(j)) >= 0 for j in E.J])
    KV = np.zeros((8, 64, 4))
    for tile in range(SN // 16):
        xin = E.to_tiles(X[16 * tile: 16 * tile + 16], E.COL_NAT88, 6, 88)
        kt, vt = E.gemm(blob, ik, xin, swap=True), E.gemm(blob, iv, xin, swap=True)
        for h in range(8):
            for r in range(4):
                kk = np.where(slot_ok, elu1(kt[h][:, r]), 0.0)
                vv = np.where(slot_ok, vt[h][:, r] / SN, np.where(E.J == 3, 1.0, 0.0))
                KV[h] = E.mfma16(kk, vv, KV[h])
    Q = elu1(X @ Wq.T).reshape(SN, 8, 11)
    K = elu1(X @ Wk.T).reshape(SN, 8, 11)
    V = (X @ Wv.T).reshape(SN, 8, 11) / SN
    ref = np.einsum("lhd,hdv->lhv", Q, np.einsum("shd,shv->hdv", K, V)) / (np.einsum("lhd,hd->lh", Q, K.sum(0))[..., None] + 1e-6) * SN
    ref_merged = ref.reshape(SN, 88) @ Wm.T
    for tile in range(SN // 16):
        q = E.gemm(blob, iq, E.to_tiles(X[16 * tile: 16 * tile + 16], E.COL_NAT88, 6, 88))
        assert q.shape[0] == 6
        msg_tiles = np.zeros((6, 64, 4))
        for h in range(8):
            acc = np.zeros((64, 4))
            for qd in range(3):
                quad = 3 * h + qd
                qq = np.where(3 * E.G + qd < 11, elu1(q[quad >> 2][:, quad & 3]), 0.0)
                acc = E.mfma16(KV[h][:, qd], qq, acc)
            den = acc[E.J, 3]                              # slot 3: lane group 0, register 3
            msg = acc * (1.0 / (den + 1e-6) * SN)[:, None]
            for lane in range(64):
                for r in range(3):
                    v = E.head11_slot(4 * (lane >> 4) + r)
                    if v >= 0:
                        assert abs(msg[lane, r] - ref[16 * tile + (lane & 15), h, v]) < 1e-9
            for rr in range(3):
                quad = 3 * h + rr
                msg_tiles[quad >> 2][:, quad & 3] = msg[:, rr]
        merged, pad = E.from_tiles(E.gemm(blob, im, msg_tiles), E.ROW_NAT88, 88)
        assert pad == 0.0
        np.testing.assert_allclose(merged, ref_merged[16 * tile: 16 * tile + 16], rtol=1e-9, atol=1e-9)


# ------------------------------------------------------------------ fp16 plane region (split-precision MFMA path)
@pytest.fixture(scope="module")
def f16_blob(lib, raw_and_blob):
    raw, _ = raw_and_blob
    n = lib.ufr_packed_f16_halfwords()
    pid, el, pl = (np.zeros(n, np.int32) for _ in range(3))
    P32 = C.POINTER(C.c_int32)
    assert lib.ufr_pack_plan_f16(pid.ctypes.data_as(P32), el.ctypes.data_as(P32), pl.ctypes.data_as(P32)) == 0
    planes = {p: E.split2(raw[p].astype(np.float32), E.W_SCALE) for p in np.unique(pid) if p >= 0}
    blob = np.zeros(n, np.float32)
    for p, sp in planes.items():
        for k in range(2):
            sel = (pid == p) & (pl == k)
            blob[sel] = sp[k].reshape(-1)[el[sel]]
    return blob


def test_f16_split_carries_22_bits():
    """hi + lo of the two-plane fp16 split: within 2^-22 of the value over the supported range (measured max ~2^-23),
    both planes fp16 numbers; tiny values degrade gradually (absolute floor 2^-25 / scale)."""
    rng = np.random.default_rng(1)
    x = (rng.standard_normal(20000) * np.exp(rng.uniform(-6, 5, 20000))).astype(np.float32)
    x = x[np.abs(x) < 4000]
    h, l = E.split2(x, E.X_SCALE)
    err = np.abs((h.astype(np.float64) + l) / 16.0 - x)
    assert (err <= np.maximum(np.abs(x) * 2.0 ** -22, 2.0 ** -29)).all()
    for p in (h, l):
        assert np.array_equal(p.astype(np.float16).astype(np.float32), p)


def test_f16_region_size(lib):
    assert lib.ufr_packed_f16_halfwords() == E.f16_region_frags() * 512


@pytest.mark.parametrize("name", [m[0] for m in E.MATS])
def test_fp16x3_panels_reproduce_linear(name, raw_and_blob, f16_blob):
    """The exported plane plan, pushed through a lane-level model of v_mfma_f32_16x16x32_f16 with the three
    plane pairs of csrc/weight_stream_f16.h, reproduces y = W x to fp32 accuracy for every matrix of the chain."""
    raw, _ = raw_and_blob
    idx = E.NAME2IDX[name]
    _, param, k_raw, n_out, n_in, rm, cm, out_dim, in_dim = E.MATS[idx]
    W = raw[param].reshape(out_dim, k_raw).astype(np.float32)
    x = np.random.default_rng(200 + idx).standard_normal((16, in_dim)).astype(np.float32)
    tiles = E.to_tiles(x, cm, n_in, in_dim).astype(np.float32)
    ref = x.astype(np.float64) @ W[:, :in_dim].astype(np.float64).T
    if name in ("RT_K", "RT_V"):  # swapped operands in the kernel: [token][feature] accumulators
        y, pad = E.from_tiles_swapped(E.gemm_f16(f16_blob, name, tiles, swap=True), rm, out_dim), 0.0
    else:
        y, pad = E.from_tiles(E.gemm_f16(f16_blob, name, tiles), rm, out_dim)
    assert pad == 0.0
    assert np.abs(y - ref).max() / np.abs(ref).max() < 5e-7, name


# ------------------------------------------------------------------ bf16 plane region (backward data-gradient chains)
def _bf16(x):
    """round to nearest even to bf16, returned as float32"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def _split_bf16(x):
    hi = _bf16(x)
    return hi, _bf16(np.asarray(x, np.float32) - hi)


# (name, param, raw shape, n_out, n_in, row map, col map, out_dim, in_dim, k-steps) in the order of stream B_VTB
BWD_MATS = [("RW2T", 34, (8, 16), 1, 1, E.ROW_NAT, E.COL_NAT, 16, 8), ("RW0T", 32, (16, 83), 5, 1, E.ROW_NAT, E.COL_NAT, 80, 16),
            ("MLP2T", 11, (80, 160), 10, 5, E.ROW_NAT, E.COL_NAT, 160, 80), ("MLP0T", 10, (160, 160), 10, 10, E.ROW_NAT, E.COL_NAT, 160, 160),
            ("MERGET", 9, (80, 80), 5, 5, E.ROW_SLOT20, E.COL_NAT, 80, 80), ("QT", 6, (80, 80), 5, 5, E.ROW_NAT, E.COL_SLOT20, 80, 80),
            ("KT", 7, (80, 80), 5, 5, E.ROW_NAT, E.COL_SLOT20, 80, 80), ("VT", 8, (80, 80), 5, 5, E.ROW_NAT, E.COL_SLOT20, 80, 80)]


@pytest.fixture(scope="module")
def bwd_blob(lib, raw_and_blob):
    raw, _ = raw_and_blob
    n = lib.ufr_packed_bwd_halfwords()
    pid, el, pl = (np.zeros(n, np.int32) for _ in range(3))
    P32 = C.POINTER(C.c_int32)
    assert lib.ufr_pack_plan_bwd(pid.ctypes.data_as(P32), el.ctypes.data_as(P32), pl.ctypes.data_as(P32)) == 0
    blob = np.zeros(n, np.float32)
    for p in np.unique(pid):
        if p < 0:
            continue
        planes = _split_bf16(raw[p].astype(np.float32).reshape(-1))
        for k in range(2):
            sel = (pid == p) & (pl == k)
            blob[sel] = planes[k][el[sel]]
    return blob


@pytest.mark.parametrize("mi", range(len(BWD_MATS)))
def test_bf16x3_transposed_panels_reproduce_the_data_gradient(mi, raw_and_blob, bwd_blob):
    """The exported plan of the backward region, pushed through the lane-level MFMA model with bf16 hi/lo planes and the
    three plane products of csrc/weight_stream_f16.h, reproduces d in = W^T d out for every matrix of the view
    transformer's data-gradient chain (csrc/view_dgrad.hip) to 16-bit-split accuracy."""
    raw, _ = raw_and_blob
    f0 = 0
    for name, param, shape, n_out, n_in, rm, cm, out_dim, in_dim in BWD_MATS[:mi]:
        f0 += ((n_in + 1) // 2) * n_out * 2
    name, param, shape, n_out, n_in, rm, cm, out_dim, in_dim = BWD_MATS[mi]
    W = raw[param].reshape(shape).astype(np.float32)
    dy = np.random.default_rng(300 + mi).standard_normal((16, in_dim)).astype(np.float32)
    tiles = E.to_tiles(dy, cm, n_in, in_dim).astype(np.float32)
    ref = dy.astype(np.float64) @ W.astype(np.float64)[:in_dim, :out_dim]        # (16, out_dim) = dY W
    out = np.zeros((n_out, 64, 4), np.float32)
    zero = np.zeros((64, 4), np.float32)
    for s in range((n_in + 1) // 2):
        ta, tb = tiles[2 * s], (tiles[2 * s + 1] if 2 * s + 1 < n_in else zero)
        xb = [np.concatenate([pa, pb], axis=1) for pa, pb in zip(_split_bf16(ta), _split_bf16(tb))]
        for to in range(n_out):
            f = f0 + (s * n_out + to) * 2
            a = [bwd_blob[(f + p) * 512:(f + p + 1) * 512].reshape(64, 8) for p in range(2)]
            for pa, pb in ((1, 0), (0, 1), (0, 0)):
                out[to] = E.mfma_f16(a[pa], xb[pb], out[to])
    y, pad = E.from_tiles(out, rm, out_dim)
    assert pad == 0.0
    assert np.abs(y - ref).max() / np.abs(ref).max() < 3e-5, name


def test_mfma_identity_transposes_a_tile_into_contraction_layout():
    """csrc/wgrad_stream.hip: a stored tile (lane (g, j): features 4g..4g+3 of token j) fed as the A operand of the 16x16x32
    MFMA against the selector B[8g + i][n] = (i < 4 and 4g + i == n) comes out as lane (g', j') = feature j', tokens
    4g'..4g'+3 -- and two such operands contract to sum_t dY[o][t] X[i][t]."""
    rng = np.random.default_rng(7)
    dY, X = (_bf16(rng.standard_normal((2, 16, 16))) for _ in range(2))          # [column tile][feature][token]
    sel = np.zeros((64, 8), np.float32)
    for lane in range(64):
        g, j = lane >> 4, lane & 15
        if (j >> 2) == g:
            sel[lane, j & 3] = 1.0

    def frag(V):
        parts = []
        for c in range(2):
            a = np.zeros((64, 8), np.float32)
            for lane in range(64):
                g, j = lane >> 4, lane & 15
                a[lane, :4] = V[c, 4 * g:4 * g + 4, j]
            parts.append(E.mfma_f16(a, sel, np.zeros((64, 4), np.float32)))
        for lane in range(64):                                                 # lane = feature j', registers = tokens 4g' + r
            g, j = lane >> 4, lane & 15
            for c in range(2):
                assert np.array_equal(parts[c][lane], V[c, j, 4 * g:4 * g + 4])
        return np.concatenate(parts, axis=1)

    acc = E.mfma_f16(frag(dY), frag(X), np.zeros((64, 4), np.float32))
    ref = np.einsum("cot,cit->oi", dY.astype(np.float64), X.astype(np.float64))
    for lane in range(64):
        g, j = lane >> 4, lane & 15
        np.testing.assert_allclose(acc[lane], ref[4 * g:4 * g + 4, j], rtol=1e-6, atol=1e-6)


def test_header_is_usable_from_plain_c(lib, tmp_path):
    """include/ufr.h compiled as C by gcc, linked against libufr.so: every entry point resolves and the argument checks
    answer without a GPU (tests/cabi/abi_check.c)."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc on this host")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "uforecon_amd", "lib")
    exe = str(tmp_path / "abi_check")
    import re

    header = open(os.path.join(root, "include", "ufr.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)                       # declarations only, not the prose around them
    names = sorted(set(re.findall(r"\b(ufr_[a-z0-9_]+)\s*\(", header)))
    assert len(names) >= 69 and "ufr_conv3d_bwd_weight_heads" in names and "ufr_conv3d_planes" in names and "ufr_conv3d_bwd_weight" in names and "ufr_weights_fit_frame" in names
    (tmp_path / "abi_syms.inc").write_text(",\n".join(f"(const void*){n}" for n in names) + "\n")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), "-I", str(tmp_path),
                    os.path.join(root, "tests", "cabi", "abi_check.c"), "-L", libdir, "-lufr", f"-Wl,-rpath,{libdir}",
                    "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert f"abi ok: {len(names)} entry points" in r.stdout
