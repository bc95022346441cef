"""LDS layouts against the gfx950 bank model (tools/lds_bank_model.py): the address arithmetic of the kernels is restated here lane by lane, and every fragment /
tap read that round 3's census found conflicting (profiles/r03v_pmc_census_full.txt) must now cost the conflict-free cycle count - for every tap, wave and pass.
The restatements follow conv_patch.hip (`lds_piece_a`, `fetch`), mbconv.hip (phase 2), encoder_head.hip (`ss_off`, `ds_off`, the split image planes) and
masking.hip (`maxpool_sq_lds_kernel`, x pass); a layout change in a kernel has to be mirrored here, which is the point."""
import pytest

from tools.lds_bank_model import read_b128_cycles, read_b32_cycles, write_b128_cycles


def test_the_model_reproduces_the_measured_conflicts_of_the_round2_layouts():
    # halo stage swizzled by the LINEAR pixel index: every A read 2-way (8 cycles) -> 16 of 40 LDS cycles per k-step = the measured 36-39 %
    def a_old(lane, wrow=3, ky=1, kx=1, sstep=0):
        fr, hb = lane & 31, lane >> 5
        pix = (wrow + (fr >> 4) + ky) * 18 + (fr & 15) + kx
        return pix * 32 + (((2 * (2 * sstep + hb)) ^ ((pix >> 1) & 7)) << 2)
    assert read_b128_cycles(a_old) == 8
    # fused MBConv, stride 1, E row stride 36 floats: depthwise tap reads 2-way
    assert read_b128_cycles(lambda l: ((l >> 3) + 11) * 36 + (l & 7) * 4) == 8
    # encoder head, [pixel][16 floats] unswizzled: 4-way at stride 1, 8-way at stride 2
    assert read_b128_cycles(lambda l: ((l // 16) * 18 + l % 16) * 16 + 4) == 16
    assert read_b128_cycles(lambda l: ((2 * (l // 8)) * 17 + 2 * (l % 8)) * 16 + 4) == 32


@pytest.mark.parametrize("wrow", range(16))
def test_patch_kernel_halo_stage_reads_are_conflict_free(wrow):
    """conv_patch.hip: lane (fr, hb) reads 16-byte piece pc of halo pixel (row, rx + kx); swizzle by the COLUMN rx + kx."""
    for ky in range(3):
        for kx in range(3):
            for sstep in range(2):
                for lo in range(2):
                    def addr(lane):
                        fr, hb = lane & 31, lane >> 5
                        ry, rx = fr >> 4, fr & 15
                        pix, x = (wrow + ry + ky) * 18 + rx + kx, rx + kx
                        pc = 2 * (2 * sstep + hb) + lo
                        return pix * 32 + ((pc ^ ((x >> 1) & 7)) << 2)
                    assert read_b128_cycles(addr) == 4, (wrow, ky, kx, sstep, lo)
    # the weight rows keep the row-index swizzle (32 consecutive rows per fragment)
    for pc in range(8):
        assert read_b128_cycles(lambda lane: (lane & 31) * 32 + (((pc & 6) + 2 * 0 ^ (((lane & 31) >> 1) & 7)) << 2)) == 4


def test_mbconv_stride1_depthwise_reads_are_conflict_free():
    """mbconv.hip phase 2: lane = (pixel p = tid >> 3, channel quad c4 = tid & 7), E row stride MB_ES(1) = 32 floats, halo row width 10."""
    ES, WI, TWO = 32, 10, 8
    for wave in range(4):
        for q in range(2):
            for ky in range(3):
                for kx in range(3):
                    def addr(lane):
                        tid = 64 * wave + lane
                        p = (tid >> 3) + 32 * q
                        oy, ox = p // TWO, p % TWO
                        return ((oy + ky) * WI + ox + kx) * ES + (tid & 7) * 4
                    assert read_b128_cycles(addr) == 4


def _ss_off(q, hx, quad):
    return q * 16 + ((quad ^ ((hx >> 1) & 3)) << 2)


def _ds_off(p, quad):
    return p * 16 + ((quad ^ ((p + (p >> 2)) & 3)) << 2)


def test_encoder_head_layouts():
    # stride 1: depthwise tap reads of Ss (TS = 18, TO = 16) conflict-free for every quad, pass and tap; stride 2 (TS = 17, TO = 8): 2-way is this layout's floor
    for S, TS, TO, want in ((1, 18, 16, 4), (2, 17, 8, 8)):
        for quad in range(4):
            for it in range(max(1, TO * TO // 64)):
                for ky in range(3):
                    for kx in range(3):
                        def addr(lane):
                            p = lane + 64 * it
                            y, x = p // TO, p % TO
                            return _ss_off((y * S + ky) * TS + x * S + kx, x * S + kx, quad)
                        assert read_b128_cycles(addr) == want
    # Ds: pointwise reads (64 consecutive pixels per wave, quad c4) and depthwise writes conflict-free
    for base in (0, 64, 128, 192):
        for c4 in range(4):
            assert read_b128_cycles(lambda lane: _ds_off(base + lane, c4)) == 4
            assert write_b128_cycles(lambda lane: _ds_off(base + lane, c4)) == 8
    # stem reads of the split image planes (stride-1 variant: TS = 18, row pitch 41, odd plane at +21): conflict-free across the row wraps
    TS, TIP, HALF = 18, 41, 21
    for j in range(3):
        for w in range(2):
            for ky in range(3):
                for kx in range(3):
                    def addr(lane):
                        p = min(64 * w + lane + 128 * j, TS * TS - 1)
                        y, x = p // TS, p % TS
                        return 2 * y * TIP + x + ky * TIP + (kx & 1) * HALF + (kx >> 1)
                    assert read_b32_cycles(addr) == 2


def test_maxfilter_window_reads_are_vectors():
    """masking.hip x pass: lane i reads its 4 + 2R window as 16-byte vectors at a lane stride of 4 floats; the row stride keeps every vector inside its row."""
    for W, R in ((224, 10), (224, 5), (112, 5)):
        SW = (W + 3) // 4 * 4 + (2 * R + 3) // 4 * 4 + 4
        NV = (4 + 2 * R + 3) // 4
        W4 = (W + 3) // 4
        assert SW % 4 == 0 and (W4 * 4 - 4) + 4 * NV <= SW
        for j in range(NV):
            # inside a row: consecutive 16-byte vectors, conflict-free; a wave that wraps into the next row (56 vectors per 224-wide row) pays 2-4 extra cycles
            assert read_b128_cycles(lambda lane: lane * 4 + 4 * j if lane < min(W4, 64) else None) == 4
            assert read_b128_cycles(lambda lane: (lane // W4) * SW + (lane % W4) * 4 + 4 * j) <= 8      # (the four scalar reads it replaces: 4 x 8 cycles)
            # the scalar form it replaces: one float per lane at a stride of 4 floats = 4-way
            assert read_b32_cycles(lambda lane: lane * 4 + j) == 8


def test_implicit_gemm_row_swizzle_is_conflict_free_for_consecutive_rows():
    """conv.hip / conv_pp.hip / conv_halo.hip: 128-byte rows, piece ^ ((row >> 1) & 7).  A fragment read takes 32 CONSECUTIVE rows (raster-order GEMM rows + a tap
    shift); the swizzle is a bijection on the real lane groups for every starting row, which is why these kernels measure 1-4 % conflict cycles (the remainder: tap reads
    that cross an image row, where the 32 rows are not consecutive)."""
    for base in range(64):
        for pc in range(8):
            def addr(lane):
                row = base + (lane & 31)
                return row * 32 + (((pc ^ (2 * (lane >> 5))) ^ ((row >> 1) & 7)) << 2)
            assert read_b128_cycles(addr) == 4

    def hs_b_off(row, stage, piece):                                 # conv_halo.hip: B ring, blocks of 8 rows with the three stages of a block adjacent
        return (((row >> 3) * 3 + stage) << 8) + ((row & 7) << 5) + ((piece ^ ((row >> 1) & 7)) << 2)
    for stage in range(3):
        for pc in range(8):
            assert read_b128_cycles(lambda lane: hs_b_off(lane & 31, stage, pc ^ (2 * (lane >> 5)))) == 4
