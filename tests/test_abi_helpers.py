"""Host-side helper entry points of the C ABI (no kernel is launched, no GPU needed): the geometry answers the Python mirror
sizes its buffers with must stay consistent with what the launches use."""
import ctypes as C

import pytest


@pytest.fixture(scope="module")
def lib():
    import t2v_hip
    return t2v_hip.load_library()


def test_conv_stat_blocks_follow_the_tile_choice(lib):
    # encoder bank (6 x 84 positions): 48-position tiles with the input channels cut in two -> 2 tiles per item
    assert lib.t2v_conv1d_stat_blocks(6, 84, 512, 512, 5) == 12
    # Postnet (6 x 400): 80-position tiles, no split
    assert lib.t2v_conv1d_stat_blocks(6, 400, 512, 512, 5) == 30
    # generic kernel (Cin not a multiple of 16): 64 output positions per block over the flattened batch
    assert lib.t2v_conv1d_stat_blocks(2, 129, 33, 70, 3) == (2 * 129 + 63) // 64


def test_reverse_pass_geometry(lib):
    assert lib.t2v_decoder_bwd_persist_slices(84) == 6 and lib.t2v_decoder_bwd_persist_slices(224) == 7
    assert lib.t2v_decoder_bwd_persist_supported(7, 84) == 0 and lib.t2v_decoder_bwd_achain_dq_offset(7, 84, 400) == -1
    if lib.t2v_decoder_bwd_persist_supported(6, 84) == 1:        # asks the device for its CU count: 0 on a machine without one
        off = lib.t2v_decoder_bwd_achain_dq_offset(6, 84, 400)
        total = lib.t2v_decoder_bwd_achain_scratch_floats(6, 84, 400)
        assert off > 0 and off % 4 == 0 and off + 400 * 6 * 128 <= total      # the summed-dq rows lie inside the scratch
    else:
        assert lib.t2v_decoder_bwd_achain_dq_offset(6, 84, 400) == -1


def test_scratch_size_answers_are_monotonic(lib):
    assert lib.t2v_gemm_splitk_scratch_floats(81, 1536, 2400) > 0            # the projection's weight gradient is split
    # large products: no scratch for the fp32-MFMA kernels; the x3 path (round 6) keeps the bf16 planes of both operands there
    # (6 bytes per element, padded to whole tiles) + the raw tiles of its k-splits
    prev = lib.t2v_gemm_f32_set_mode(0)
    try:
        assert lib.t2v_gemm_splitk_scratch_floats(4096, 2560, 2400) == 0
        lib.t2v_gemm_f32_set_mode(1)
        planes = 6 * (4096 + 2560) * 2400 // 4
        got = lib.t2v_gemm_splitk_scratch_floats(4096, 2560, 2400)
        assert planes <= got <= planes + 4 * 4096 * 2560
        assert lib.t2v_gemm_splitk_scratch_floats(81, 1536, 2400) > 0 and lib.t2v_gemm_splitk_scratch_floats(128, 128, 2400) > 0   # small: fp32 split-K
    finally:
        lib.t2v_gemm_f32_set_mode(prev)
    assert lib.t2v_colsum_scratch_floats(2400, 4096) >= 4096 and lib.t2v_colsum_scratch_floats(1, 7) == 0
    a, b = lib.t2v_decoder_train_persist_scratch_floats(6, 84, 100), lib.t2v_decoder_train_persist_scratch_floats(6, 84, 400)
    assert 0 < a < b
