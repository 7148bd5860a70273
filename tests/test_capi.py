"""The C-ABI shared library builds (hipcc cross-compiles gfx950 without a GPU), loads, and exports
every symbol include/gps_hip.h declares.  No compute calls here."""
import ctypes
import os
import re

from sceneverse_amd import _native


def test_library_exports_every_declared_symbol():
    lib = _native.load()
    declared = _native.declared_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/gps_hip.h but not exported"
    assert set(_native.SIGNATURES) <= set(declared)
    assert lib.gps_abi_version() == 11
    assert lib.gps_error_string(0) == b"ok"
    assert b"not supported" in lib.gps_error_string(-2)


def test_header_cites_reference_interfaces():
    text = open(_native.HEADER_PATH).read()
    for cite in ("src/sampling.cpp", "src/ball_query.cpp", "src/group_points.cpp", "src/interpolate.cpp",
                 "bindings.cpp:6-19"):
        assert cite in text


def test_argument_validation_without_gpu():
    """Pure host-side checks of the C ABI: bad sizes are rejected before any launch, empty work is OK."""
    lib = _native.load()
    assert lib.gps_ball_query(-1, 4, 4, 0.1, 4, None, None, None, None) == -1
    assert lib.gps_group_points(0, 3, 8, 2, 2, None, None, None, None) == 0
    assert lib.gps_furthest_point_sampling(0, 8, 4, None, None, None, None) == 0
    assert lib.gps_three_nn(1, 0, 5, None, None, None, None, None) == 0
    # fused entry points: empty batches are fine, unsupported shapes say so, bad arguments are caught
    assert lib.gps_sa_mlp_forward(0, 32, 16, 32, 128, 128, 128, 256, None, None, None, None, None, None, None) == 0
    assert lib.gps_sa_mlp_forward(1, 32, 16, 16, 128, 128, 128, 256, 1, 1, 1, 1, 1, 1, None) == -2   # nsample != 32
    assert lib.gps_sa_mlp_forward_bf16x3(1, 32, 16, 32, 7, 64, 64, 64, 1, 1, 1, 1, 1, 1, None) == -2  # widths
    assert lib.gps_sa_mlp_layer_floats(6, 64) == 2 * 512 and lib.gps_sa_mlp_layer_floats(6, 65) == -1
    assert lib.gps_sa_mlp_wpack_floats(131, 128, 128, 256) == 16 * 4352
    assert lib.gps_attn_forward(0, 12, 80, 64, None, None, None, 2304, None, None, None, 0.0, 0, None, None, 768,
                                None, None) == 0
    assert lib.gps_attn_forward(2, 12, 80, 32, 1, 1, 1, 2304, None, None, None, 0.0, 0, None, 1, 768, 1, None) == -2
    assert lib.gps_attn_forward(2, 12, 80, 64, 1, 1, 1, 2304, 1, None, None, 0.0, 0, None, 1, 768, 1, None) == -1
    assert lib.gps_attn_forward(2, 12, 600, 64, 1, 1, 1, 2304, None, None, None, 0.0, 0, None, 1, 768, 1, None) == -2
    assert lib.gps_masked_ce_forward(0, 30522, 1, None, 30522, None, -1, None, None, None) == 0
    assert lib.gps_masked_ce_forward(4, 30522, 1, 1, 100, 1, -1, 1, 1, None) == -1                   # ld < vocab
    assert lib.gps_masked_ce_forward_rows(0, 30522, 1, None, 30528, None, -1, None, None, None, None, None, None) == 0
    assert lib.gps_masked_ce_backward_rows(4, 30522, 1, 1, 30528, 1, -1, None, 1, 1, None, None, 1, 100, None) == -1   # ldd < vocab
    assert lib.gps_masked_ce_forward_rows(4, 30522, 1, 16, 30528, 1, -1, None, 1, 1, 1, None, None) == -1      # mean without ticket
    assert lib.gps_lm_row_plan(8, 0, 1, -1, 1, 1, 1, None) == -1                                     # vocab < 1
    assert lib.gps_lm_row_plan(8, 10, None, -1, 1, 1, 1, None) == -1
    assert lib.gps_cloud_compact(4, 0, 6, 16, 16, 1, 1, 1, 1, 16, 16, None) == -1                     # n < 1
    assert lib.gps_cloud_compact(4, 1024, 2, 16, 16, 1, 1, 1, 1, 16, 16, None) == -1                  # ld < 3
    assert lib.gps_cloud_compact(4, 1024, 6, None, 16, 1, 1, 1, 1, 16, 16, None) == -1                # no cloud
    lib.gps_point_set_object_extent(None)                                                             # (a no-op reset)
    assert lib.gps_text_obj_ce_forward(0, 80, 768, None, None, None, None, 1e-12, -100, None, None, None, None, None, None,
                                       None, None) == 0
    assert lib.gps_text_obj_ce_forward(4, 80, 770, 16, 16, 1, 1, 1e-12, -100, 1, 1, 1, 1, 1, 1, 1, None) == -2   # D % 4
    assert lib.gps_text_obj_ce_backward(4, 80, 768, 16, 16, 1, 1e-12, -100, 1, 1, 1, 1, 1, None, 16, 16, None) == -1
    assert lib.gps_clip_loss_forward(0, 768, 1, 16, 16, 1, 100.0, 1e-12, 1, 1, 1, 1, 1, 1, 1, 1, None) == -1     # n < 1
    assert lib.gps_clip_loss_backward(4, 768, 1, 16, 16, 1, 100.0, 1e-12, 1, 1, 1, 1, 1, 1, 16, None, 1, 1, 1, None) == -1
    assert lib.gps_add_dropout_layernorm_forward(4, 100, 0, 1, 1, 1, 1, 1, 1e-5, 0.0, 0, None, 1, None, 1, 1,
                                                 None) == -2                                          # width
    assert lib.gps_add_dropout_layernorm_forward(0, 768, 0, 1, None, None, None, None, 1e-5, 0.0, 0, None, None,
                                                 None, None, None, None) == 0
    assert lib.gps_ln_partial_rows(8320) == 1024 and lib.gps_ln_partial_rows(5) == 2
    # gps_embedding_grad(n, d, num_rows, ids, dy, ld, padding_idx, scratch, out, stream)
    assert lib.gps_embedding_grad(4, 768, 0, None, None, 768, -1, None, None, None) == 0
    assert lib.gps_embedding_grad(4, 770, 10, 1, 16, 772, -1, 1, 16, None) == -2    # d not a multiple of 4
    assert lib.gps_embedding_grad(4, 4096, 10, 1, 16, 4096, -1, 1, 16, None) == -2  # d > 2048
    assert lib.gps_embedding_grad(4, 768, 10, 1, 16, 100, -1, 1, 16, None) == -1    # ld < d
    assert lib.gps_embedding_grad(-1, 768, 10, 1, 16, 768, -1, 1, 16, None) == -1
    assert lib.gps_colsum_bf16(4, 768, 1, 100, 1, 1, None) == -1                   # ld < cols
    assert lib.gps_colsum_bf16(4, 100, 16, 104, 1, 1, None) == -2                  # cols not a multiple of 8
    assert lib.gps_colsum_bf16(4, 0, None, 0, None, None, None) == 0
    assert lib.gps_colsum_parts(19200, 768) == 240 and lib.gps_colsum_parts(0, 768) == 0
    # gps_obj_processing_post(n_rows, n_points, xyz, rgb, rgb_is_u8, offsets, row_obj, sample_idx, seed, rot,
    #                         row_rot, fts, locs, boxes, masks, stream)
    assert lib.gps_obj_processing_post(0, 1024, None, None, 1, None, None, None, 0, None, None, None, None, None,
                                       None, None) == 0
    assert lib.gps_obj_processing_post(4, 0, 1, 1, 1, 1, 1, None, 0, None, None, 1, 1, None, None, None) == -1
    assert lib.gps_obj_processing_post(4, 4096, 1, 1, 1, 1, 1, None, 0, None, None, 1, 1, None, None, None) == -2
    assert lib.gps_obj_processing_post(4, 1024, 1, None, 1, 1, 1, None, 0, None, None, 1, 1, None, None, None) == -1
    assert lib.gps_obj_processing_post(4, 1024, 1, 1, 1, 1, 1, None, 0, 1, None, 1, 1, None, None, None) == -1  # rot w/o row_rot


def test_no_oracle_import_in_product_code():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle\.", re.M)
    for dirpath, _, files in os.walk(os.path.join(root, "sceneverse_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                code = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))
                code = re.sub(r'""".*?"""', "", code, flags=re.S)
                assert not pat.search(code), f"{f} references oracle/"


def test_library_has_no_undefined_kernel_symbols():
    """Every kernel stub the launch code references is defined in the library itself (hipcc's host pass can
    silently drop the stubs of a kernel template whose body it cannot parse: that shows up only as an
    undefined symbol when the library is loaded with immediate binding on the GPU box)."""
    import subprocess
    from sceneverse_amd import _native
    _native.load()
    out = subprocess.run(["nm", "-D", "--undefined-only", _native.LIB_PATH], capture_output=True, text=True).stdout
    bad = [l for l in out.splitlines() if "gps_" in l]
    assert not bad, bad[:5]
