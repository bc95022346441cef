"""Host-side model of the split-fp16 weight-gradient kernels' data movement (smirk_amd/csrc/train.hip: wgrad_f16_kernel, wgrad3x3_halo_f16_kernel).

The kernels stage split16 operands into LDS with `slot(h, g, k)` and build their MFMA operands with gfx950's transpose read `ds_read_b64_tr_b16`.  This file
restates (a) the slot function, (b) the transpose read as measured on the MI355X (tools/tr_probe.py, profiles/r02w_wgrad_f16_sweep.txt: lane 4r+q of a 16-lane
group supplies pixel r / channels 4q..4q+3, lane c receives the four pixels of channel c), (c) the operand layout of v_mfma_f32_32x32x16_f16, and checks that
the per-lane byte offsets the kernels use reproduce dW = dZ^T X exactly for every tile shape and every halo instantiation, and that the layout is bank-conflict
free for the transpose reads and the staging writes.  It is the index arithmetic that is tested, in numpy; the kernels themselves are tested on the GPU
(tests/test_train_ops_gpu.py::test_conv_weight_gradient)."""
import numpy as np
import pytest


def wgf_slot(ngq, kq_per, h, g, k):
    return ((h * ngq + (g >> 2)) * kq_per + (k >> 2)) * 16 + ((((k & 3) + (g >> 2)) & 3) << 2) + (g & 3)


def tr_read(lds, byte_addr):
    """ds_read_b64_tr_b16: per 16-lane group a [4][16] block of halves; lane i = 4r+q supplies row r, columns 4q..4q+3; lane c receives column c"""
    out = np.zeros((64, 4), dtype=lds.dtype)
    for grp in range(4):
        blk = np.zeros((4, 16), dtype=lds.dtype)
        for i in range(16):
            a = int(byte_addr[grp * 16 + i])
            assert a % 8 == 0
            blk[i >> 2, 4 * (i & 3):4 * (i & 3) + 4] = lds[a // 2:a // 2 + 4]
        out[grp * 16:grp * 16 + 16] = blk.T
    return out


def frag(lds, off):
    return np.concatenate([tr_read(lds, off), tr_read(lds, off + 256)], axis=1)            # [64 lanes][8 k]


def mfma_32x32x16(A, B):
    """A, B [64][8]: lane l holds row/column l % 32, k = 8 * (l // 32) + e"""
    Am, Bm = np.zeros((32, 16)), np.zeros((16, 32))
    for l in range(64):
        Am[l % 32, 8 * (l // 32):8 * (l // 32) + 8] = A[l]
        Bm[8 * (l // 32):8 * (l // 32) + 8, l % 32] = B[l]
    return Am @ Bm


LANES = np.arange(64)
LI = LANES & 15
R4, Q4 = LI >> 2, LI & 3
GL, KHALF = ((LANES >> 4) & 1) * 2 + (Q4 >> 1), LANES >> 5


def lane_off(gq, kq_per, k_first):
    """byte offset of (half hi, first read) for the 32-channel quad gq; k_first = this lane's first pixel (8 * khalf + r4 + tap shift)"""
    return (((gq * kq_per + (k_first >> 2)) * 16) + ((((k_first & 3) + gq) & 3) << 2) + GL) * 16 + (Q4 & 1) * 8


@pytest.mark.parametrize("TM,Cout", [(128, 128), (64, 64), (32, 32), (128, 72), (64, 40)])
def test_generic_tile_offsets_reproduce_the_gemm(TM, Cout):
    rng = np.random.default_rng(TM + Cout)
    GA, AQ, BQ = TM // 8, TM // 32, 4
    dz = rng.integers(-3, 4, size=(16, Cout)).astype(np.float64)
    x = rng.integers(-3, 4, size=(16, 128)).astype(np.float64)
    A, B = np.zeros(GA * 2 * 16 * 8), np.zeros(16 * 2 * 16 * 8)                           # chunk images in halves
    for tid in range(256):                                                                # the staging threads: (pixel sk, group sg)
        sk, sg = tid >> 4, tid & 15
        if sg < GA and sg * 8 < Cout:
            s = wgf_slot(AQ, 4, 0, sg, sk) * 8
            A[s:s + 8] = dz[sk, sg * 8:sg * 8 + 8]
        s = wgf_slot(BQ, 4, 0, sg, sk) * 8
        B[s:s + 8] = x[sk, sg * 8:sg * 8 + 8]
    WAVES_M = 2 if TM == 128 else 1
    WAVES_N = 4 // WAVES_M
    BM, BN = TM // 32 // WAVES_M, 4 // WAVES_N
    dW = np.zeros((TM, 128))
    for wave in range(4):
        wm, wn = wave // WAVES_N, wave % WAVES_N
        for i in range(BM):
            fa = frag(A, lane_off(wm * BM + i, 4, KHALF * 8 + R4))
            for j in range(BN):
                fb = frag(B, lane_off(wn * BN + j, 4, KHALF * 8 + R4))
                dW[(wm * BM + i) * 32:(wm * BM + i) * 32 + 32, (wn * BN + j) * 32:(wn * BN + j) * 32 + 32] = mfma_32x32x16(fa, fb)
    ref = np.zeros((TM, 128))
    ref[:min(TM, Cout)] = (dz.T @ x)[:min(TM, Cout)]
    assert np.array_equal(dW, ref)


@pytest.mark.parametrize("TM,CIN,NW", [(32, 32, 3), (64, 32, 6), (32, 64, 6), (64, 64, 12)])
def test_halo_offsets_serve_all_nine_taps_from_one_staged_halo(TM, CIN, NW):
    rng = np.random.default_rng(TM * CIN)
    MB, NBQ, GA, GB, HQ = TM // 32, CIN // 32, TM // 8, CIN // 8, 14
    NB = MB * 9 * NBQ
    dz = rng.integers(-3, 4, size=(16, TM)).astype(np.float64)
    xh = rng.integers(-3, 4, size=(3, 18, CIN)).astype(np.float64)
    A, B = np.zeros(GA * 2 * 16 * 8), np.zeros(GB * 2 * HQ * 4 * 8)
    for e in range(16 * GA):
        k, g = e // GA, e % GA
        s = wgf_slot(GA // 4, 4, 0, g, k) * 8
        A[s:s + 8] = dz[k, g * 8:g * 8 + 8]
    for e in range(54 * GB):
        hp, g = e // GB, e % GB
        s = wgf_slot(GB // 4, HQ, 0, g, hp) * 8
        B[s:s + 8] = xh[hp // 18, hp % 18, g * 8:g * 8 + 8]
    dW, done = np.zeros((TM, 9 * CIN)), np.zeros(NB, bool)
    for wave in range(NW):
        fas = [frag(A, lane_off(m, 4, KHALF * 8 + R4)) for m in range(MB)]
        for i in range((NB + NW - 1) // NW):
            blk = wave + NW * i
            if blk >= NB:
                continue
            mb, rem = blk // (9 * NBQ), blk % (9 * NBQ)
            tap, nbq = rem // NBQ, rem % NBQ
            fb = frag(B, lane_off(nbq, HQ, (tap // 3) * 18 + (tap % 3) + KHALF * 8 + R4))
            dW[mb * 32:mb * 32 + 32, tap * CIN + nbq * 32:tap * CIN + nbq * 32 + 32] = mfma_32x32x16(fas[mb], fb)
            done[blk] = True
    assert done.all() and len({(NB + NW - 1 - w) // NW for w in range(NW)}) == 1        # every block computed once, every wave owns the same number
    ref = np.zeros_like(dW)
    for tap in range(9):
        ref[:, tap * CIN:(tap + 1) * CIN] = dz.T @ xh[tap // 3, tap % 3:tap % 3 + 16, :]
    assert np.array_equal(dW, ref)


def test_layout_is_bank_conflict_free():
    """transpose reads: a half-wave (32 lanes x 8 bytes) must touch 64 distinct 4-byte banks (bank = (addr / 4) % 64); staging writes (ds_write_b128, 8 lanes
    = 8 consecutive groups of one pixel per LDS cycle, bank = (addr / 4) % 32): 8 distinct 16-byte bank slots.  (SQ_LDS_BANK_CONFLICT measured 0,
    profiles/r02ac_pmc_wgrad_f16.txt.)"""
    for gq in range(4):
        for kshift in range(0, 12):                                                       # the halo kernel shifts the pixel index by the tap offset
            off = lane_off(gq, 14, kshift + R4)[:32]
            banks = np.concatenate([(off // 4) % 64, (off // 4 + 1) % 64])
            assert len(set(banks.tolist())) == 64, (gq, kshift)
    for g0 in (0, 8):
        for k in range(16):
            assert len({wgf_slot(4, 4, 0, g0 + g, k) % 8 for g in range(8)}) == 8
