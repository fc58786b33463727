"""Version-string parser vs the reference's substring precedence (davo.py:1010-1450)."""
import pytest

from davo_amd.version import (parse_version, weight_shapes, FLAGSHIP_VERSION, UnsupportedVariantError)


def test_flagship():
    c = parse_version(FLAGSHIP_VERSION)
    assert (c.major, c.use_flow_info, c.cnv6_out, c.se_act) == ("v1", True, 128, "tanh")
    assert (c.norm_flow, c.abs_mode, c.att_source, c.mask_rgb, c.mask_info) == (False, "all", "se_flow", True, True)
    assert c.cin_per_frame == 5
    sh = weight_shapes(c)
    assert len(sh) == 26                                    # SURVEY table W
    assert sum(int.__mul__(*(lambda s: (1, __import__("math").prod(s)))(s)) for s in sh.values()) == 1590361
    assert sh["pose_exp_net/cnv1/weights"] == (7, 7, 10, 16)
    assert sh["pose_exp_net/pose/translation/pred/weights"] == (1, 1, 256, 3)


def test_abs_flow_precedence():
    # -abs_flow_h / -abs_flow_v are tested before -abs_flow (davo.py:1094-1102)
    base = "v1-sharedNN-dilatedPoseNN-segmask_all-se_flow"
    assert parse_version(base + "-abs_flow_h").abs_mode == "h"
    assert parse_version(base + "-abs_flow_v").abs_mode == "v"
    assert parse_version(base + "-abs_flow").abs_mode == "all"
    assert parse_version(base).abs_mode == "none"
    assert parse_version(base + "-norm_flow-abs_flow").norm_flow


def test_activation_and_cnv6():
    base = "v1-sharedNN-dilatedPoseNN-segmask_all-se_flow"
    assert parse_version(base).se_act == "relu"            # davo.py:1083-1085 default
    assert parse_version(base + "-fc_lrelu").se_act == "lrelu"
    assert parse_version(base + "-cnv6_64").cnv6_out == 64
    assert parse_version(base).cnv6_out == 128             # davo.py:1053 default


def test_masking_modes():
    assert parse_version("v1-sharedNN-dilatedPoseNN-segmask_rgb-se_flow").mask_info is False
    assert parse_version("v1-sharedNN-dilatedPoseNN-segmask_rgb-se_flow").mask_rgb is True
    c = parse_version("v1-sharedNN-dilatedPoseNN-se_flow")  # no -segmask_: attention built but unused
    assert (c.mask_rgb, c.mask_info) == (False, False)
    c = parse_version("v0-sharedNN-dilatedPoseNN-segmask-se_flow")
    assert (c.use_flow_info, c.cin_per_frame, c.mask_rgb, c.mask_info) == (False, 3, True, False)
    # a version with no leading vN is "v0" (davo.py:1056-1057)
    assert parse_version("davo-sharedNN-dilatedPoseNN-no_segmask").major == "v0"


def test_attention_source_order():
    assert parse_version("v1-sharedNN-dilatedPoseNN-no_segmask").att_source == "ones"
    assert parse_version("v1-sharedNN-dilatedPoseNN-segmask_all-static").att_source == "static_src"
    assert parse_version("v1-sharedNN-dilatedPoseNN-segmask_all").att_source == "static_all"
    # -se_flow wins over -no_segmask because it comes first in the elif chain (davo.py:1175 vs 1385)
    assert parse_version("v1-sharedNN-dilatedPoseNN-no_segmask-se_flow").att_source == "se_flow"


@pytest.mark.parametrize("v", [
    "v1-sharedNN-dilatedPoseNN-segmask_all-se_flow_on_depthseg",      # precedes -se_flow (davo.py:1156)
    "v1-sharedNN-dilatedPoseNN-se_gp2x2_flow",
    "v1-sharedNN-dilatedPoseNN-se_spp_flow",
    "v1-sharedNN-dilatedPoseNN-se_depth",
    "v1-sharedNN-dilatedPoseNN-se_seg",
    "v1-sharedNN-dilatedPoseNN-se_flow-se_insert",
    "v1-sharedNN-dilatedPoseNN-se_flow-batch_norm",
    "v1-sharedNN-dilatedPoseNN-se_flow-seglabelid",
    "v1-sharedNN-dilatedCouplePoseNN-se_flow",
    "v1-dilatedPoseNN-se_flow",                                       # non-shared nets
    "v1-se_flow",
    "v1.555-sharedNN-dilatedPoseNN-segmask_all-se_flow",
    "v1-sharedNN-dilatedPoseNN-se_flow-cnv6_96",
])
def test_unsupported_raise_nameerror(v):
    with pytest.raises(NameError):                          # reference raises NameError (davo.py:1035-1037)
        parse_version(v)
    with pytest.raises(UnsupportedVariantError):
        parse_version(v)


def test_reference_nameerrors():
    with pytest.raises(NameError, match="not support `-sharedNN-couplePoseNN' mode."):
        parse_version("v1-sharedNN-couplePoseNN")
    with pytest.raises(NameError, match="unknown PoseNN type."):
        parse_version("v1-sharedNN")


def test_none_version_asserts():
    with pytest.raises(AssertionError):                     # davo.py:959
        parse_version(None)
