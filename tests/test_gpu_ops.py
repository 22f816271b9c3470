"""Projection kernels (GEMV / MFMA GEMM / generic GEMM) through fl_op_linear vs numpy."""
import numpy as np
import pytest

import synth
from conftest import needs_experimental
from conftest import experimental_build

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import fastllm_amd
    assert fastllm_amd.device_count() >= 1
    return fastllm_amd


def _rand(shape, seed, scale=1.0):
    return (np.random.RandomState(seed).standard_normal(shape) * scale).astype(np.float32)


def _ref(x, w, bias, epi):
    y = x.astype(np.float64) @ w.astype(np.float64).T
    if bias is not None:
        y = y + bias
    if epi:
        I = w.shape[0] // 2
        g, u = y[:, :I], y[:, I:]
        y = g / (1.0 + np.exp(-g)) * u
    return y


# (T, N, K): GEMV (T=1), MFMA (T>1, K%64==0), generic (K%64!=0), ragged M/N tails, big K
SHAPES = [(1, 512, 256), (1, 6144, 4096), (1, 300, 384), (1, 4096, 14336), (1, 2, 8), (1, 33, 1032),
          (5, 512, 256), (128, 256, 512), (130, 384, 448), (257, 1000, 1024), (7, 96, 40), (512, 512, 4096), (300, 1024, 2048), (512, 4096, 4096), (129, 2048, 5632),
          (1030, 26000, 256), (520, 44000, 192), (2050, 16400, 128),   # 256x128 three-stage kernel (>= 1024 tiles)
          (64, 136, 72),
          # decode GEMV with a partial last 64-lane chunk after 0..3 full ones (tensor-parallel K slices: Mistral down_proj
          # at tp 8, Qwen2 down_proj / o_proj at tp 4) and K below one chunk row
          (1, 4096, 1792), (1, 3584, 4736), (1, 3584, 896), (1, 256, 2568), (1, 64, 520), (1, 40, 24),
          # two row tiles whose 256x128 grid with K slices is exactly one round of the chip (op_linear allows 4 slabs)
          (512, 4096, 14336), (500, 4096, 4096)]


@pytest.mark.parametrize("T,N,K", SHAPES)
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_linear_plain(fa, T, N, K, dtype):
    x, w, b = _rand((T, K), 1), _rand((N, K), 2, 0.05), _rand((N,), 3)
    if dtype == "bf16":
        xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
        y = fa.op_linear(xb, wb, b)
        ref = _ref(synth.bf16_bits_to_f32(xb), synth.bf16_bits_to_f32(wb), b, 0)
    else:
        y = fa.op_linear(x, w, b)
        ref = _ref(x, w, b, 0)
    tol = 2e-5 * np.sqrt(K) + 1e-5           # fp32 accumulation of K products of O(0.05)
    np.testing.assert_allclose(y, ref, atol=tol, rtol=1e-5)


@pytest.mark.parametrize("T,N,K", [(1, 6144, 4096), (1, 4096, 14336), (1, 3584, 4736), (8, 4096, 4096), (64, 2048, 5632), (128, 11264, 2048),
                                   (512, 6144, 4096), (512, 4096, 14336), (512, 28672, 4096), (1100, 4096, 4096), (2050, 16400, 128),
                                   (4100, 4608, 3584), (4096, 3584, 2048)])
def test_linear_bit_exact_on_small_integers(fa, T, N, K):
    """Production bf16 kernels (stream GEMV, short-prompt GEMM, 256x256 with K slices / peeled stream-K tails, 256x128, 128x128)
    on integer-valued operands: every product and every partial sum is an integer below 2^24, so fp32 accumulation is EXACT in
    any order and the result must equal the integer reference bit for bit -- a tiling, indexing or K-slice error cannot hide in a
    tolerance."""
    rs = np.random.RandomState(T * 31 + N * 7 + K)
    x = rs.randint(-3, 4, size=(T, K)).astype(np.float32)
    w = rs.randint(-3, 4, size=(N, K)).astype(np.float32)
    assert K * 9 < 2 ** 24
    y = fa.op_linear(synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w), None)
    ref = x.astype(np.float64) @ w.astype(np.float64).T       # integers below 2^53: the BLAS product is exact in any order too
    assert np.array_equal(ref, np.rint(ref)) and np.abs(ref).max() < 2 ** 24
    np.testing.assert_array_equal(y, ref.astype(np.float32))


@pytest.mark.parametrize("T,I,K", [(1, 352, 256), (1, 14336, 4096), (1, 40, 64), (9, 352, 256), (200, 704, 512),
                                   (128, 1792, 1024), (3, 24, 48), (1025, 13000, 192)])
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_linear_silu_gate(fa, T, I, K, dtype):
    """gate/up rows in HF order in, silu(gate)*up out (the library interleaves them 16x16 itself)."""
    x, w = _rand((T, K), 4), _rand((2 * I, K), 5, 0.05)
    if dtype == "bf16":
        xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
        y = fa.op_linear(xb, wb, None, epilogue=1)
        ref = _ref(synth.bf16_bits_to_f32(xb), synth.bf16_bits_to_f32(wb), None, 1)
        # output is stored as bf16
        np.testing.assert_allclose(y, ref, atol=1e-3, rtol=2 ** -8)
    else:
        y = fa.op_linear(x, w, None, epilogue=1)
        ref = _ref(x, w, None, 1)
        np.testing.assert_allclose(y, ref, atol=2e-5 * np.sqrt(K) + 1e-5, rtol=1e-5)


def test_mfma_operand_maps_with_asymmetric_data(fa):
    """A = I-like / asymmetric B check (cdna guide section 3): catches swapped C/D or k-order maps."""
    T, N, K = 128, 128, 64
    x = np.zeros((T, K), np.float32)
    x[np.arange(T), np.arange(T) % K] = 1.0                     # row t selects column t%K
    w = (np.arange(N)[:, None] * 3 + np.arange(K)[None, :] * 7 % 11).astype(np.float32)   # asymmetric, exact in bf16? keep small
    w = (w % 64).astype(np.float32)
    y = fa.op_linear(synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w))
    ref = x @ w.T
    np.testing.assert_array_equal(y, ref)


# 256x256 phase-interleaved GEMM (k_gemm_8p.hip): forced on shapes with ragged M / N tails, a single K pair, split-K
# slabs, bias and the SiLU-gate epilogue; and picked by the default heuristic on a chip-filling grid
@pytest.mark.parametrize("T,N,K,epi,bias", [(256, 256, 128, 0, True), (300, 700, 256, 0, True), (513, 1000, 1024, 0, False),
                                            (512, 4096, 4096, 0, False), (700, 1300, 192, 1, False), (260, 4096, 640, 1, False)])
@pytest.mark.parametrize("force", ["2", None])
@pytest.mark.parametrize("four", ["0", "2"])                  # the eight-wave kernel / its four-wave form (default: from 768 tokens)
def test_linear_8phase_kernel(fa, T, N, K, epi, bias, force, four, monkeypatch):
    monkeypatch.setenv("FL_GEMM_4W", four)
    if force:
        monkeypatch.setenv("FL_GEMM_8P", force)
    x, w = _rand((T, K), 11), _rand((N, K), 12, 0.05)
    b = _rand((N,), 13) if bias else None
    xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
    y = fa.op_linear(xb, wb, b, epilogue=epi)
    ref = _ref(synth.bf16_bits_to_f32(xb), synth.bf16_bits_to_f32(wb), b, epi)
    if epi:
        np.testing.assert_allclose(y, ref, atol=1e-3, rtol=2 ** -8)
    else:
        np.testing.assert_allclose(y, ref, atol=2e-5 * np.sqrt(K) + 1e-5, rtol=1e-5)


@pytest.mark.parametrize("four", ["0", "2"])
def test_8phase_kernel_is_race_free_over_repeats(fa, monkeypatch, four):
    """The LDS-DMA hand-offs are ordered only by counted waits and barriers: a misplaced one shows up as rare
    wrong tiles.  Same inputs 30 times (different workgroup timing each launch): bit-identical outputs."""
    monkeypatch.setenv("FL_GEMM_8P", "2")
    monkeypatch.setenv("FL_GEMM_4W", four)
    T, N, K = 1024, 2048, 2048
    xb, wb = synth.f32_to_bf16_bits(_rand((T, K), 21)), synth.f32_to_bf16_bits(_rand((N, K), 22, 0.05))
    first = fa.op_linear(xb, wb, None)
    ref = _ref(synth.bf16_bits_to_f32(xb), synth.bf16_bits_to_f32(wb), None, 0)
    np.testing.assert_allclose(first, ref, atol=2e-5 * np.sqrt(K) + 1e-5, rtol=1e-5)
    for _ in range(30):
        np.testing.assert_array_equal(fa.op_linear(xb, wb, None), first)


# short-prompt GEMM (k_gemm_skinny.hip): every BM bucket (32/64/128/256), ragged T and N, K slices, bias, SiLU gate
@pytest.mark.parametrize("T,N,K,epi,bias", [(2, 4096, 4096, 0, False), (17, 1000, 1024, 0, True), (33, 6144, 512, 0, True),
                                            (64, 4096, 2048, 0, False), (65, 200, 256, 0, True), (129, 1000, 1024, 0, False),
                                            (200, 4096, 4096, 0, False), (256, 3000, 768, 0, True),
                                            (9, 1792, 1024, 1, False), (48, 704, 512, 1, False), (100, 5632, 2048, 1, False),
                                            (256, 1408, 1024, 1, False)])
@pytest.mark.parametrize("loaders", [0, 1])      # 1: the staging moved to dedicated loader waves (64 / 128-token workgroups)
def test_linear_skinny_kernel(fa, monkeypatch, T, N, K, epi, bias, loaders):
    if loaders and not experimental_build():
        pytest.skip("loader-wave form: EXPERIMENTAL build only")
    monkeypatch.setenv("FL_SKINNY_LOADERS", str(loaders))
    x, w = _rand((T, K), 31), _rand((N if not epi else 2 * N, K), 32, 0.05)
    b = _rand((N,), 33) if bias else None
    xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
    y = fa.op_linear(xb, wb, b, epilogue=epi)
    ref = _ref(synth.bf16_bits_to_f32(xb), synth.bf16_bits_to_f32(wb), b, epi)
    if epi:
        np.testing.assert_allclose(y, ref, atol=1e-3, rtol=2 ** -8)
    else:
        np.testing.assert_allclose(y, ref, atol=2e-5 * np.sqrt(K) + 1e-5, rtol=1e-5)
    for _ in range(5):                                            # counted-wait pipeline: same bits every launch
        np.testing.assert_array_equal(fa.op_linear(xb, wb, b, epilogue=epi), y)


# batched-decode wide projection on the LDS-DMA ring kernel (k_gemv_dma.hip), routed through fl_op_linear by
# FL_OP_LINEAR_DMA=1: every K-tiles-per-wave bucket (4/8/12/16), K slices, ragged last unit, B = 1..8, SiLU gate;
# a wave-private ring ordered only by counted waits: same bits on every launch
@pytest.mark.parametrize("T,N,K,epi", [(8, 4096, 4096, 0), (3, 1000, 256, 0), (5, 2000, 1536, 0), (8, 512, 7168, 0),
                                       (1, 320, 512, 0), (8, 4096, 14336, 0), (7, 33, 64, 0),
                                       (8, 14336, 4096, 1), (4, 352, 256, 1), (6, 5632, 2048, 1), (2, 16, 128, 1),
                                       # round 5: 9-16 rows (the whole MFMA tile) and 17-32 (two row blocks per weight fragment)
                                       (16, 4096, 4096, 0), (11, 1000, 1536, 0), (16, 14336, 4096, 1), (9, 352, 256, 1), (13, 512, 7168, 0),
                                       (32, 4096, 4096, 0), (17, 1000, 1024, 0), (32, 14336, 4096, 1), (24, 5632, 2048, 1), (29, 9472, 3584, 1),
                                       (32, 300, 512, 0)])
def test_linear_dma_ring_kernel(fa, T, N, K, epi, monkeypatch):
    monkeypatch.setenv("FL_OP_LINEAR_DMA", "1")
    x, w = _rand((T, K), 41), _rand((N if not epi else 2 * N, K), 42, 0.05)
    xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
    y = fa.op_linear(xb, wb, None, epilogue=epi)
    ref = _ref(synth.bf16_bits_to_f32(xb), synth.bf16_bits_to_f32(wb), None, epi)
    if epi:
        np.testing.assert_allclose(y, ref, atol=1e-3, rtol=2 ** -8)
    else:
        np.testing.assert_allclose(y, ref, atol=2e-5 * np.sqrt(K) + 1e-5, rtol=1e-5)
    for _ in range(8):
        np.testing.assert_array_equal(fa.op_linear(xb, wb, None, epilogue=epi), y)
    monkeypatch.delenv("FL_OP_LINEAR_DMA")
    tight = fa.op_linear(xb, wb, None, epilogue=epi)              # the other kernels on the same data: same result up to summation order
    # (gate/up outputs are bf16 on both sides: each may sit one ulp off the exact value, in opposite directions)
    np.testing.assert_allclose(y, tight, atol=(2e-3 if epi else 2e-5 * np.sqrt(K) + 1e-5), rtol=2 ** -7 if epi else 1e-5)


# column-peeled GEMM (k_gemm_mfma.hip launch_gemm_mfma): whole rounds of 256x256 tiles on the phase-interleaved kernel, the
# remaining columns (<= a quarter round) as a second launch on smaller tiles writing its own column range of the same output
@pytest.mark.parametrize("T,N,K,epi,bias", [(4096, 4352, 1024, 0, True), (2048, 8448, 1024, 0, False), (4096, 2176, 1024, 1, False),
                                            (4000, 4300, 1088, 0, True),
                                            # tails that run on the 128 x 256 kernel with their K slices summed in the launch (Qwen2-7B's
                                            # gate/up at 512 tokens, its QKV / gate/up tails at 4096): bias / SiLU gate behind the sum
                                            (512, 37888, 1024, 0, True), (512, 18944, 1024, 1, False), (4096, 4608, 2048, 0, True)])
def test_linear_peeled_columns(fa, T, N, K, epi, bias):
    x, w = _rand((T, K), 51), _rand((N if not epi else 2 * N, K), 52, 0.05)
    b = _rand((N,), 53) if bias else None
    xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
    y = fa.op_linear(xb, wb, b, epilogue=epi)
    xf, wf = synth.bf16_bits_to_f32(xb), synth.bf16_bits_to_f32(wb)
    ref = xf @ wf.T                                               # fp32 BLAS: 1e-5-level agreement is all that is asked
    if b is not None:
        ref = ref + b
    if epi:
        g, u = ref[:, :N].astype(np.float64), ref[:, N:].astype(np.float64)
        ref = g / (1.0 + np.exp(-g)) * u
        np.testing.assert_allclose(y, ref, atol=2e-3, rtol=2 ** -7)
    else:
        np.testing.assert_allclose(y, ref, atol=4e-5 * np.sqrt(K) + 1e-4, rtol=1e-4)


# stream-K form of the 256x256 kernel (one workgroup per CU over the (tile, K step) line; pieces of a tile meet through fp32
# partials and a per-tile ticket).  The product uses it for the peeled tail of a long prompt's GEMM (above); FL_GEMM_STREAMK=3
# runs whole matrices through it: aligned pieces, pieces that straddle tiles (41 K steps cut in two), ragged edges, gate/up
@pytest.mark.parametrize("T,N,K,epi,bias", [(2048, 3584, 4096, 0, False), (1024, 6144, 2624, 0, True), (2000, 3000, 1088, 0, True),
                                            (1024, 2176, 2624, 1, False), (300, 520, 512, 0, False),
                                            (1400, 1500, 1088, 0, True)])   # six row tiles: a group of four and a group of two (the four-wave kernel's tile order)
@pytest.mark.parametrize("four", ["0", "2"])
def test_linear_streamk_whole(fa, monkeypatch, T, N, K, epi, bias, four):
    monkeypatch.setenv("FL_GEMM_STREAMK", "3")
    monkeypatch.setenv("FL_GEMM_4W", four)
    x, w = _rand((T, K), 61), _rand((N if not epi else 2 * N, K), 62, 0.05)
    b = _rand((N,), 63) if bias else None
    xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
    y = fa.op_linear(xb, wb, b, epilogue=epi)
    y2 = fa.op_linear(xb, wb, b, epilogue=epi)                    # tickets are back at zero, partial slots are reused
    assert np.array_equal(y, y2)
    ref = synth.bf16_bits_to_f32(xb) @ synth.bf16_bits_to_f32(wb).T
    if b is not None:
        ref = ref + b
    if epi:
        g, u = ref[:, :N].astype(np.float64), ref[:, N:].astype(np.float64)
        np.testing.assert_allclose(y, g / (1.0 + np.exp(-g)) * u, atol=2e-3, rtol=2 ** -7)
    else:
        np.testing.assert_allclose(y, ref, atol=4e-5 * np.sqrt(K) + 1e-4, rtol=1e-4)


# the 256x256 kernel below one full row tile and just past a tile boundary (the cost model sends e.g. a 255- or 257-token gate/up
# there): rows past T are clamped on load and skipped on store
@pytest.mark.parametrize("T,N,K,epi,bias", [(130, 4096, 1024, 0, True), (200, 2048, 2048, 1, False), (255, 4608, 1024, 0, False),
                                            (257, 3072, 1024, 0, True), (257, 1536, 2048, 1, False)])
@pytest.mark.parametrize("four", ["0", "2"])
def test_linear_8phase_ragged_rows(fa, monkeypatch, T, N, K, epi, bias, four):
    monkeypatch.setenv("FL_GEMM_8P", "2")
    monkeypatch.setenv("FL_GEMM_4W", four)
    monkeypatch.setenv("FL_GEMM_SKINNY", "0")
    x, w = _rand((T, K), 71), _rand((N if not epi else 2 * N, K), 72, 0.05)
    b = _rand((N,), 73) if bias else None
    xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
    y = fa.op_linear(xb, wb, b, epilogue=epi)
    ref = _ref(synth.bf16_bits_to_f32(xb), synth.bf16_bits_to_f32(wb), b, epi)
    if epi:
        np.testing.assert_allclose(y, ref, atol=2e-3, rtol=2 ** -7)
    else:
        np.testing.assert_allclose(y, ref, atol=2e-5 * np.sqrt(K) + 1e-5, rtol=1e-5)


# 128 x 256 GEMM whose K slices meet inside the launch (k_gemm_h4.hip): forced on (gemm_h4 = 2) with 1-4 slices.  Integer operands:
# the sum of the slices is exact in any order, so a wrong block, a missed slice or a stale partial shows as a wrong integer.
# wait_us = 0 makes every early slice abandon its blocks at once, so that the slice counted last finishes them from memory
# (the rescue path, which ordinary timing never takes).
@pytest.mark.parametrize("T,N,K", [(128, 256, 64), (130, 300, 512), (512, 4096, 4096), (257, 1000, 1024), (300, 520, 6400), (640, 4096, 2048)])
@pytest.mark.parametrize("slices", [1, 2, 3, 4])
@pytest.mark.parametrize("wait_us", [30, 0])
def test_linear_h4_slices_meet_in_the_launch(fa, T, N, K, slices, wait_us):
    if K // 64 < slices or (slices == 1 and wait_us == 0):
        pytest.skip("no such case")
    rs = np.random.RandomState(T + N + K)
    x = rs.randint(-3, 4, size=(T, K)).astype(np.float32)
    w = rs.randint(-3, 4, size=(N, K)).astype(np.float32)
    ref = (x.astype(np.float64) @ w.astype(np.float64).T).astype(np.float32)
    xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
    try:
        fa.tune("gemm_h4", 2); fa.tune("h4_split", slices); fa.tune("h4_wait_us", wait_us)
        y = fa.op_linear(xb, wb, None)
        np.testing.assert_array_equal(y, ref)
        for _ in range(6):                                        # the tile words alternate between two sets: several launches in a row
            np.testing.assert_array_equal(fa.op_linear(xb, wb, None), y)
    finally:
        fa.tune("reload_env", 0)


# fp32 mode's prompt GEMM on v_mfma_f32_32x32x2_f32 (k_gemm_f32.hip): exact fp32 products summed as an fmaf chain in k order -- bit for
# bit the 64 x 64 VALU kernel (gemm_f32_mfma = 0) on ragged T and N, both tile heights (64 / 128 rows), one and many K tiles
@pytest.mark.parametrize("T,N,K,epi,bias", [(2, 100, 32, 0, True), (64, 4096, 4096, 0, False), (300, 1000, 1040, 0, True), (130, 6144, 16, 0, False),
                                            (512, 8192, 512, 1, False), (77, 352, 256, 1, False), (1025, 13000, 192, 1, False), (2100, 4608, 3584, 0, True)])
def test_linear_f32_mfma_is_the_fmaf_chain(fa, T, N, K, epi, bias):
    x, w = _rand((T, K), 101), _rand((N if not epi else 2 * N, K), 102, 0.05)
    b = _rand((N,), 103) if bias else None
    try:
        fa.tune("gemm_f32_mfma", 1)
        y = fa.op_linear(x, w, b, epilogue=epi)
        fa.tune("gemm_f32_mfma", 0)
        y0 = fa.op_linear(x, w, b, epilogue=epi)
    finally:
        fa.tune("reload_env", 0)
    np.testing.assert_array_equal(y, y0)
    np.testing.assert_allclose(y, _ref(x, w, b, epi), atol=2e-5 * np.sqrt(K) + 1e-5, rtol=1e-5)


# 256 x 224 four-wave GEMM (k_gemm_w14.hip), forced on (gemm_w14 = 2): widths of whole 224-column tiles, two / odd / many K tiles,
# ragged last row tiles.  Integer operands: any wrong fragment, column block or stale LDS half shows as a wrong integer.
@pytest.mark.parametrize("T,N,K", [(512, 448, 128), (257, 224, 192), (300, 2240, 1024), (1024, 896, 4096), (700, 28672, 256)])
def test_linear_w14_tiles_of_224_columns(fa, T, N, K):
    rs = np.random.RandomState(T + N + K)
    x = rs.randint(-3, 4, size=(T, K)).astype(np.float32)
    w = rs.randint(-3, 4, size=(N, K)).astype(np.float32)
    ref = (x.astype(np.float64) @ w.astype(np.float64).T).astype(np.float32)
    xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
    try:
        fa.tune("gemm_h4", 0); fa.tune("gemm_w14", 2)
        y = fa.op_linear(xb, wb, None)
        np.testing.assert_array_equal(y, ref)
        for _ in range(3):
            np.testing.assert_array_equal(fa.op_linear(xb, wb, None), y)
    finally:
        fa.tune("reload_env", 0)


@pytest.mark.parametrize("T,N,K,epi,bias", [(512, 448, 512, 0, True), (300, 560, 1024, 1, False), (640, 1120, 512, 1, False), (1000, 672, 320, 0, True)])
def test_linear_w14_epilogues(fa, T, N, K, epi, bias):
    """bias / the SiLU-gate epilogue on the 224-column tile (its last column group is two blocks wide: one gate/up pair)."""
    x, w = _rand((T, K), 91), _rand((N if not epi else 2 * N, K), 92, 0.05)
    b = _rand((N,), 93) if bias else None
    xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
    ref = _ref(synth.bf16_bits_to_f32(xb), synth.bf16_bits_to_f32(wb), b, epi)
    try:
        fa.tune("gemm_h4", 0); fa.tune("gemm_w14", 2)
        y = fa.op_linear(xb, wb, b, epilogue=epi)
        fa.tune("gemm_w14", 0)
        y0 = fa.op_linear(xb, wb, b, epilogue=epi)
    finally:
        fa.tune("reload_env", 0)
    if epi:
        np.testing.assert_allclose(y, ref, atol=2e-3, rtol=2 ** -7)
    else:
        np.testing.assert_allclose(y, ref, atol=2e-5 * np.sqrt(K) + 1e-5, rtol=1e-5)
    # same K order and the same epilogue arithmetic as the 256-column kernels: identical fp32 results (a gate/up output may differ
    # from a kernel with another accumulation pattern by one bf16 rounding)
    if epi:
        np.testing.assert_allclose(y, y0, atol=1e-6, rtol=2 ** -7)
    else:
        np.testing.assert_array_equal(y, y0)


@pytest.mark.parametrize("T,N,K,epi,bias", [(2, 4096, 4096, 0, True), (5, 1000, 512, 0, False), (16, 6144, 4096, 0, True), (17, 520, 1280, 0, True),
                                            (40, 4096, 14336, 0, False), (64, 300, 256, 0, True), (3, 352, 256, 1, False), (16, 14336, 4096, 1, False),
                                            (33, 1024, 2304, 1, False), (9, 2, 256, 0, True)])
def test_linear_f32_weight_stream_for_few_rows(fa, T, N, K, epi, bias):
    """Round 5: fp32 projections of 2-64 token rows (short prompts and decode batches of the fp32 mode) as a weight stream -- 16 weight
    rows and all token rows of a 16-row block per workgroup, x in LDS (gemv_f32_rows_kernel) -- against fp64 numpy, and against the
    MFMA tiles it replaces there (another order of the fp32 sums)."""
    x, w = _rand((T, K), 301), _rand((N if not epi else 2 * N, K), 302, 0.05)
    b = _rand((N,), 303) if bias else None
    ref = _ref(x, w, b, epi)
    try:
        y = fa.op_linear(x, w, b, epilogue=epi)
        fa.tune("f32_rows_max", 0)
        y0 = fa.op_linear(x, w, b, epilogue=epi)
    finally:
        fa.tune("reload_env", 0)
    tol = 2e-5 * np.sqrt(K) + 1e-5
    np.testing.assert_allclose(y, ref, atol=tol, rtol=1e-5)
    np.testing.assert_allclose(y, y0, atol=tol, rtol=1e-5)


def _fuzz_shapes(seed, n):
    """Seeded projection shapes biased to the planner's edges: token counts around tile heights and policy thresholds, widths around
    224 / 256-column tiles and chip-filling counts, K around the K-tile and slice boundaries."""
    rs = np.random.RandomState(seed)
    edges_t = [1, 2, 8, 16, 17, 32, 33, 64, 65, 96, 128, 129, 176, 192, 256, 257, 384, 512, 513, 545, 608, 609, 640, 768, 769, 1024, 1025, 1121]
    out = []
    while len(out) < n:
        T = int(edges_t[rs.randint(len(edges_t))] + rs.randint(-1, 2)) if rs.rand() < 0.7 else int(rs.randint(1, 1300))
        T = max(T, 1)
        epi = int(rs.rand() < 0.35)
        N = int(rs.choice([224, 256, 448, 1792, 2048, 3584, 4096, 6144, 7168, 8960, 11264, 14336, 28672])) if rs.rand() < 0.6 else int(rs.randint(2, 900)) * 32
        if rs.rand() < 0.25 and not epi:
            N += int(rs.randint(1, 31))                       # ragged widths (plain epilogue only: gate/up rows come in groups of 32)
        K = int(rs.choice([64, 128, 256, 512, 1024, 2048, 3584, 4096, 5632])) if rs.rand() < 0.6 else int(rs.randint(2, 700)) * 8
        if (T * N * K > 1.2e10) or (N * K > 6e7):
            continue
        out.append((T, N, K, epi, bool(rs.rand() < 0.3) and not epi))
    return out


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_linear_shape_fuzz_bf16(fa, seed):
    """Every shape goes to SOME kernel (launch_linear's plans: GEMV, ring kernel, short-prompt GEMM, 128 x 256 / 256 x 224 / 256 x 256
    tiles, K slices, peeled columns, stream-K tails, the gate/up row split): 14 seeded shapes per seed, biased to the planner's edges,
    each against fp64 numpy.  A hole between two plans (a shape nobody takes correctly) shows here."""
    for T, N, K, epi, bias in _fuzz_shapes(seed, 14):
        x, w = _rand((T, K), 1000 + seed), _rand((N, K), 2000 + seed, 0.05)
        b = _rand((N,), 3000 + seed) if bias else None
        xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
        ref = _ref(synth.bf16_bits_to_f32(xb), synth.bf16_bits_to_f32(wb), b, epi)
        y = fa.op_linear(xb, wb, b, epilogue=epi)
        msg = "T=%d N=%d K=%d epi=%d bias=%s" % (T, N, K, epi, bias)
        if epi:
            np.testing.assert_allclose(y, ref, atol=2e-3, rtol=2 ** -7, err_msg=msg)
        else:
            np.testing.assert_allclose(y, ref, atol=2e-5 * np.sqrt(K) + 1e-5, rtol=1e-5, err_msg=msg)


@pytest.mark.parametrize("seed", [11, 12])
def test_linear_shape_fuzz_f32(fa, seed):
    """The fp32 mode's plans (GEMV, the weight stream for 2-64 rows, MFMA tiles, the generic kernel) over seeded shapes."""
    for T, N, K, epi, bias in _fuzz_shapes(seed, 12):
        if T * N * K > 3e9:
            T = max(1, int(3e9 // (N * K)))
        x, w = _rand((T, K), 1000 + seed), _rand((N, K), 2000 + seed, 0.05)
        b = _rand((N,), 3000 + seed) if bias else None
        ref = _ref(x, w, b, epi)
        y = fa.op_linear(x, w, b, epilogue=epi)
        np.testing.assert_allclose(y, ref, atol=2e-5 * np.sqrt(K) + 1e-5, rtol=1e-5, err_msg="T=%d N=%d K=%d epi=%d bias=%s" % (T, N, K, epi, bias))


@pytest.mark.parametrize("T", [513, 530, 545, 577, 608, 1030, 1120])
def test_linear_gate_up_row_split_past_an_even_tile_count(fa, T):
    """Round 5: gate/up of a prompt 1 ... 96 tokens past an EVEN number of 256-row tiles -- the even part on the 224-column kernel (whole
    rounds of the chip), the last rows as a launch of their own (gemv / ring kernel / short-prompt GEMM, by row count):
    against fp64 numpy, and against the one-launch form (another accumulation pattern: one bf16 rounding apart)."""
    I, K = 4480, 320                                   # 2 I = 40 tiles of 224 columns
    x, w = _rand((T, K), 191), _rand((2 * I, K), 192, 0.05)
    xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
    ref = _ref(synth.bf16_bits_to_f32(xb), synth.bf16_bits_to_f32(wb), None, 1)
    try:
        fa.tune("gemm_h4", 0); fa.tune("gemm_w14", 2)
        y = fa.op_linear(xb, wb, None, epilogue=1)
        fa.tune("gateup_rowsplit", 0)
        y0 = fa.op_linear(xb, wb, None, epilogue=1)
    finally:
        fa.tune("reload_env", 0)
    np.testing.assert_allclose(y, ref, atol=2e-3, rtol=2 ** -7)
    np.testing.assert_allclose(y, y0, atol=1e-6, rtol=2 ** -7)
    T0 = (T + 255) // 256 * 256 - 256
    np.testing.assert_array_equal(y[:T0], y0[:T0])      # the even part: the same kernel, the same tiles


@pytest.mark.parametrize("T,N,K,epi,bias", [(512, 4096, 4096, 0, True), (200, 1408, 1024, 1, False), (384, 704, 512, 1, False), (300, 1000, 1024, 0, True)])
@pytest.mark.parametrize("slices", [1, 2, 4])
def test_linear_h4_epilogues(fa, T, N, K, epi, bias, slices):
    """bias and the SiLU-gate epilogue behind the in-launch sum (a K-sliced launch of the older kernels could run neither)."""
    x, w = _rand((T, K), 81), _rand((N if not epi else 2 * N, K), 82, 0.05)
    b = _rand((N,), 83) if bias else None
    xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
    ref = _ref(synth.bf16_bits_to_f32(xb), synth.bf16_bits_to_f32(wb), b, epi)
    try:
        fa.tune("gemm_h4", 2); fa.tune("h4_split", slices)
        y = fa.op_linear(xb, wb, b, epilogue=epi)
    finally:
        fa.tune("reload_env", 0)
    if epi:
        np.testing.assert_allclose(y, ref, atol=2e-3, rtol=2 ** -7)
    else:
        np.testing.assert_allclose(y, ref, atol=2e-5 * np.sqrt(K) + 1e-5, rtol=1e-5)


# Short-prompt GEMM whose K slices meet inside the launch (k_gemm_skf.hip; forced for every plain shape with gemm_skf = 3): integer
# operands, so the sum of the slices is exact in any order -- a missed slice, a stale partial tile or a ticket word that was not reset
# shows as a wrong integer.  All three token tiles (32 / 64 / 128 rows), ragged T and N, a bias, repeated launches on one workspace.
@pytest.mark.parametrize("T,N,K", [(2, 64, 256), (16, 4096, 4096), (33, 1000, 1024), (64, 6144, 4096), (100, 520, 6400), (128, 4096, 14336), (17, 384, 512)])
@pytest.mark.parametrize("slices", [1, 2, 3, 4])
def test_linear_skf_slices_meet_in_the_launch(fa, T, N, K, slices):
    if K // 64 < slices:
        pytest.skip("no such case")
    rs = np.random.RandomState(T + N + K)
    x = rs.randint(-3, 4, size=(T, K)).astype(np.float32)
    w = rs.randint(-3, 4, size=(N, K)).astype(np.float32)
    b = rs.randint(-8, 9, size=(N,)).astype(np.float32) if N % 3 == 0 else None
    ref = (x.astype(np.float64) @ w.astype(np.float64).T + (b if b is not None else 0.0)).astype(np.float32)
    xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
    try:
        fa.tune("gemm_skf", 3); fa.tune("skf_split", slices)
        for _ in range(5):
            np.testing.assert_array_equal(fa.op_linear(xb, wb, b), ref)
    finally:
        fa.tune("reload_env", 0)


@needs_experimental
@pytest.mark.parametrize("T,I,K", [(5, 352, 256), (32, 14336, 4096), (96, 1024, 2048), (128, 5632, 2048)])
def test_linear_skf_gate_up(fa, T, I, K):
    """its silu(gate) * up epilogue against fp64 (gate rows [0, I), up rows [I, 2 I) in HF order: the library interleaves them)"""
    x, w = _rand((T, K), 7), _rand((2 * I, K), 8, 0.05)
    xb, wb = synth.f32_to_bf16_bits(x), synth.f32_to_bf16_bits(w)
    ref = _ref(synth.bf16_bits_to_f32(xb), synth.bf16_bits_to_f32(wb), None, 1)
    try:
        fa.tune("gemm_skf", 3); fa.tune("prefill_dma", 0)
        y = fa.op_linear(xb, wb, None, epilogue=1)
        fa.tune("gemm_skf", 0)
        y0 = fa.op_linear(xb, wb, None, epilogue=1)
    finally:
        fa.tune("reload_env", 0)
    np.testing.assert_allclose(y, ref, atol=2e-3, rtol=2 ** -7)
    np.testing.assert_array_equal(y, y0)                 # the same K order and epilogue arithmetic as the kernel it stands in for
