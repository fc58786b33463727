"""Version-string parser for the DAVO pose-inference path.

The reference has no config object: ``DAVO(version)`` is configured by substring
tests on the ``--version`` flag, and the order of the ``if/elif`` chains matters
(reference ``davo.py:1010-1017`` se mode, ``:1027-1049`` PoseNN type, ``:1052``
cnv6 width, ``:1057-1073`` extra inputs, ``:1077-1085`` SE activation,
``:1089-1102`` SE input transform, ``:1117-1400`` attention source,
``:1417-1450`` masking).  ``parse_version`` walks the same chains in the same
order and returns a :class:`VariantConfig` for the variants this build runs on
the GPU; everything else the reference recognises is rejected with a
``NameError`` subclass, the exception type the reference itself raises for an
unknown network (``davo.py:1035-1037``, ``nets/attention_module.py:88``).
"""
import re
from dataclasses import dataclass

FLAGSHIP_VERSION = ("v1-decay100k-sharedNN-dilatedPoseNN-cnv6_128-segmask_all"
                    "-se_flow-abs_flow-fc_tanh")       # doc/arch-variants.md:8

NUM_SEG_CLASSES = 19            # utils/seg_utils/labels.py:64-101 (train ids 0..18)

# enums shared with include/davo_hip.h (davo_variant)
SE_ACT = {"relu": 0, "tanh": 1, "lrelu": 2}
ABS_MODE = {"none": 0, "h": 1, "v": 2, "all": 3}
ATT_SOURCE = {"ones": 0, "se_flow": 1, "static_src": 2, "static_all": 3}
MASK_INFO = {"none": 0, "att": 1}


class UnsupportedVariantError(NameError):
    """A version substring the reference knows but this build does not run."""


@dataclass(frozen=True)
class VariantConfig:
    version: str
    major: str               # "v0" | "v1" (regex ^(v[0-9.]+), davo.py:1056-1057)
    use_flow_info: bool      # v1: concat raw flow as 2 extra channels per frame
    cnv6_out: int            # -cnv6_(\d+), default 128 (davo.py:1052-1053)
    se_act: str              # relu | tanh | lrelu  (davo.py:1077-1085)
    norm_flow: bool          # (f-0.32140523)/15.384229 before abs (davo.py:1089-1091)
    abs_mode: str            # none | h | v | all  (davo.py:1094-1102)
    att_source: str          # ones | se_flow | static_src | static_all
    mask_rgb: bool           # rgb_k *= att_k          (davo.py:1419-1423 / 1447-1450)
    mask_info: bool          # flow_k *= att_k         (davo.py:1430-1434)

    @property
    def cin_per_frame(self):
        return 5 if self.use_flow_info else 3

    def as_c_ints(self):
        """Field order of ``davo_variant`` in include/davo_hip.h."""
        return (self.cin_per_frame, self.cnv6_out, SE_ACT[self.se_act],
                int(self.norm_flow), ABS_MODE[self.abs_mode],
                ATT_SOURCE[self.att_source], int(self.mask_rgb), int(self.mask_info))


# attention branches of davo.py:1117-1384 that come BEFORE "-se_flow" in the elif
# chain: if one of these matches, the reference never reaches the se_flow branch.
_BEFORE_SE_FLOW = ("-se_flow_on_depthseg_sharedlayers", "-se_flow_on_depthseg_seplayers",
                   "-se_flow_on_depthseg", "-se_mixDepthFlow", "-se_mixDispFlow")
# branches AFTER "-se_flow" and before "-no_segmask" (all need depth / rgb / SPP / se_block)
_AFTER_SE_FLOW = ("-se_gp2x2_flow_nobottle", "-se_gp2x2_flow", "-se_spp21_flow", "-se_spp2_flow",
                  "-se_spp_flow", "-se_spp864_flow", "-se_depth_wo_tgt_to_seg", "-se_depth_to_seg",
                  "-se_depth_wo_tgt", "-se_depth", "-se_disp_wo_tgt_to_seg", "-se_disp_to_seg",
                  "-se_disp_wo_tgt", "-se_disp", "-se_rgb_wo_tgt_to_seg", "-se_rgb_to_seg",
                  "-se_rgb_wo_tgt", "-se_rgb", "-se_seg_wo_tgt", "-se_seg", "-se_gp2x2_seg",
                  "-se_spp21_seg", "-se_spp_seg_21", "-se_spp2_seg", "-se_spp_seg", "-se_spp864_seg",
                  "-se_SegFlow_to_seg_8_wo_tgt", "-se_SegFlow_to_seg_8", "-se_SegFlow_to_seg_wo_tgt",
                  "-se_SegFlow_to_seg", "-se_mixSegFlow", "-se_spp21_mixSegFlow")


def parse_version(version):
    """Return the :class:`VariantConfig` the reference graph builder would build.

    Follows ``DAVO.build_pose_test_graph_davo`` (reference ``davo.py:955-1458``).
    """
    assert version is not None                          # davo.py:959
    v = version

    # -- davo.py:960: depth inputs are files this build does not consume
    if "depth" in v or "disp" in v:
        raise UnsupportedVariantError("version `%s': depth/disp inputs are not supported." % v)

    # -- davo.py:1010-1017: se_block inside the PoseNN
    for s in ("-se_insert", "-se_skipadd", "-se_replace"):
        if s in v:
            raise UnsupportedVariantError("version `%s': `%s' PoseNN mode is not supported." % (v, s))

    # -- davo.py:1027-1049: PoseNN type
    if "-sharedNN" in v:
        if "-dilatedPoseNN" in v:
            pass                                        # decouple_sharednet_v0_dilation
        elif "-dilatedCouplePoseNN" in v:
            raise UnsupportedVariantError("version `%s': couple_sharednet_v0_dilation is not supported." % v)
        elif "-couplePoseNN" in v:
            raise NameError("not support `-sharedNN-couplePoseNN' mode.")      # davo.py:1035
        else:
            raise NameError("unknown PoseNN type.")                            # davo.py:1037
    else:
        raise UnsupportedVariantError(
            "version `%s': only the `-sharedNN-dilatedPoseNN' network is supported." % v)

    for s in ("-batch_norm", "-dropout", "-seglabelid"):
        if s in v:
            raise UnsupportedVariantError("version `%s': `%s' is not supported." % (v, s))

    # -- davo.py:1052-1053
    m = re.search("-cnv6_([0-9]+)", v)
    cnv6_out = 128 if m is None else int(m.group(1))
    if cnv6_out not in (32, 64, 128, 256):          # the cnv7 k-order needs a power-of-two Cin
        raise UnsupportedVariantError("version `%s': cnv6 width %d is not supported." % (v, cnv6_out))

    # -- davo.py:1056-1065
    m = re.search("^(v[0-9.]+)", v)
    major = "v0" if m is None else m.group(1)
    use_flow_info = False
    if "v0" in major:
        pass
    elif "v1" in major:
        use_flow_info = True
    if ".555" in major:
        raise UnsupportedVariantError("version `%s': the `.555' masking variant is not supported." % v)

    # -- davo.py:1077-1085
    if "-fc_tanh" in v:
        se_act = "tanh"
    elif "-fc_lrelu" in v:
        se_act = "lrelu"
    else:
        se_act = "relu"

    # -- davo.py:1088-1102 (note -abs_flow_h / _v are tested before -abs_flow)
    norm_flow = "-norm_flow" in v
    if "-abs_flow_h" in v:
        abs_mode = "h"
    elif "-abs_flow_v" in v:
        abs_mode = "v"
    elif "-abs_flow" in v:
        abs_mode = "all"
    else:
        abs_mode = "none"

    # -- davo.py:1117-1400 attention source, same elif order
    for s in _BEFORE_SE_FLOW:
        if s in v:
            raise UnsupportedVariantError("version `%s': `%s' attention is not supported." % (v, s))
    if "-se_flow" in v:
        att_source = "se_flow"
    else:
        for s in _AFTER_SE_FLOW:
            if s in v:
                raise UnsupportedVariantError("version `%s': `%s' attention is not supported." % (v, s))
        if "-no_segmask" in v:
            att_source = "ones"
        elif "-segmask_" in v and "-static" in v:
            att_source = "static_src"          # davo.py:1390-1395: tgt map := ones
        else:
            att_source = "static_all"          # davo.py:1396-1400: tgt masked as well

    # -- davo.py:1415-1450 masking
    if use_flow_info:
        mask_rgb = "-segmask_" in v
        mask_info = mask_rgb and "-segmask_all" in v
    else:
        mask_rgb = "-segmask" in v
        mask_info = False

    return VariantConfig(version=v, major=major, use_flow_info=use_flow_info, cnv6_out=cnv6_out,
                         se_act=se_act, norm_flow=norm_flow, abs_mode=abs_mode,
                         att_source=att_source, mask_rgb=mask_rgb, mask_info=mask_info)


def weight_shapes(cfg):
    """TF checkpoint names -> shapes for a variant (SURVEY table W; scopes from
    nets/posenn.py:203,221-223,240, nets/attention_module.py:63,94,101,
    nets/posenn.py:386-388)."""
    c10 = 2 * cfg.cin_per_frame
    c6 = cfg.cnv6_out
    sh = {
        "pose_exp_net/cnv1/weights": (7, 7, c10, 16), "pose_exp_net/cnv1/biases": (16,),
        "pose_exp_net/cnv2/weights": (5, 5, 16, 32), "pose_exp_net/cnv2/biases": (32,),
        "pose_exp_net/cnv3/weights": (3, 3, 32, 64), "pose_exp_net/cnv3/biases": (64,),
        "pose_exp_net/cnv4/weights": (3, 3, 64, 128), "pose_exp_net/cnv4/biases": (128,),
        "pose_exp_net/cnv5/weights": (3, 3, 128, 256), "pose_exp_net/cnv5/biases": (256,),
    }
    for head in ("rotation", "translation"):
        p = "pose_exp_net/pose/%s/" % head
        sh[p + "cnv6/weights"] = (3, 3, 256, c6)
        sh[p + "cnv6/biases"] = (c6,)
        sh[p + "cnv7/weights"] = (3, 3, c6, 256)
        sh[p + "cnv7/biases"] = (256,)
        sh[p + "pred/weights"] = (1, 1, 256, 3)
        sh[p + "pred/biases"] = (3,)
    if cfg.att_source == "se_flow":
        sh["pose_exp_net/se_flow/bottleneck_fc/kernel"] = (2, 8)
        sh["pose_exp_net/se_flow/bottleneck_fc/bias"] = (8,)
        sh["pose_exp_net/se_flow/recover_fc/kernel"] = (8, NUM_SEG_CLASSES)
        sh["pose_exp_net/se_flow/recover_fc/bias"] = (NUM_SEG_CLASSES,)
    elif cfg.att_source in ("static_src", "static_all"):
        sh["pose_exp_net/pose_exp_net/seg_channel_weight/weight"] = (NUM_SEG_CLASSES,)
    return sh
