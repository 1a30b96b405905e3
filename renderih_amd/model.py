"""Drop-in for the reference's models/model.py: `HandNET_GCN(nn.Module)` + `load_model(cfg)`.

forward(img [B,3,256,256] NCHW fp32 on the GPU) -> (result, paramsDict, handDictList, otherInfo), the same
4-tuple of dicts as models/model.py:25-37; `state_dict()` keys/shapes equal the reference's (Appendix A).
"""
import os
import pickle
import torch
import torch.nn as nn

from . import assets, ops
from .config import load_cfg
from .encoder import load_encoder, flush_batches_tracked
from .decoder import decoder as Decoder


# Leave out the work of the finest mid convolution whose output `decoder.forward` drops (encoder.resnet_mid.forward,
# `drop_last`): outputs, gradients and every state_dict buffer stay bit-identical (tests/test_gpu_model.py::
# test_dead_mid_convolution_skip_changes_nothing, green on MI355X in round 4).  Default since round 4 -- measured same-box:
# training step 1771.7 -> 1773.6 images/s (inside the noise: the conv + statistics still run for the running buffers),
# fp16 inference B = 256 12137 -> 12604 images/s (+3.8 %, profiles/r04/ab/config5_*.log).  RIH_SKIP_DEAD_MID=0 computes it.
SKIP_DEAD_MID = os.environ.get('RIH_SKIP_DEAD_MID', '1') == '1'


class HandNET_GCN(nn.Module):
    def __init__(self, encoder, mid_model, decoder):
        super().__init__()
        self.encoder = encoder
        self.mid_model = mid_model
        self.decoder = decoder

    _half = None

    def use_fp16_backbone(self, enable=True):
        """Inference only (BASELINE configs[4]): run encoder + mid_model with fp16 storage and folded BatchNorm
        (renderih_amd/half.py).  Snapshots the CURRENT weights of an eval-mode model; call again after changing them.
        Used whenever the module is in eval mode under torch.no_grad(); the fp32 path is untouched otherwise."""
        if enable:
            from .half import HalfBackbone
            self._half = HalfBackbone(self.encoder, self.mid_model)
            self._half.drop_last = bool(getattr(self.decoder, 'drops_last_fmap', False))
        else:
            self._half = None
        return self

    def forward(self, img):
        with ops.owned_bounds():        # (parameter bounds measured at the top stay valid until the forward returns)
            return self._forward(img)

    def _forward(self, img):
        ops.begin_forward(self)         # operand bounds of the three-product GEMM engine are per forward pass
        if self._half is not None and not self.training and not torch.is_grad_enabled():
            hms, mask, dp, global_feature, fmaps = self._half(img)
        else:
            hms, mask, dp, img_fmaps, hms_fmaps, dp_fmaps = self.encoder(img)
            if SKIP_DEAD_MID and getattr(self.mid_model, 'supports_drop_last', False) and getattr(self.decoder, 'drops_last_fmap', False):
                global_feature, fmaps = self.mid_model(img_fmaps, hms_fmaps, dp_fmaps, drop_last=True)
            else:
                global_feature, fmaps = self.mid_model(img_fmaps, hms_fmaps, dp_fmaps)
            flush_batches_tracked()
        result, paramsDict, handDictList, otherInfo = self.decoder(global_feature, fmaps)
        if hms is not None:
            otherInfo['hms'] = hms
        if mask is not None:
            otherInfo['mask'] = mask
        if dp is not None:
            otherInfo['dense'] = dp
        return result, paramsDict, handDictList, otherInfo


Model = HandNET_GCN     # BASELINE.json's north_star calls it `models.model.Model`


def _maybe_pickle(root, rel):
    path = os.path.join(root, str(rel))
    if os.path.exists(path):
        with open(path, 'rb') as f:
            return pickle.load(f)
    return None


def load_decoder(cfg, encoder_info, asset_root=None, decoder_cls=None, extra=None):
    """models/decoder.py:177-210.  Reads the reference's misc/*.pkl when present under `asset_root` (default:
    the repo root, like the reference); otherwise the packaged graph asset + seeded synthetic stand-ins."""
    root = asset_root or os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    left = _maybe_pickle(root, cfg.MISC.GRAPH_LEFT_DICT_PATH) or assets.load_graph_dict('left')
    right = _maybe_pickle(root, cfg.MISC.GRAPH_RIGHT_DICT_PATH) or assets.load_graph_dict('right')
    dense = _maybe_pickle(root, cfg.MISC.DENSE_COLOR)
    if dense is None:
        dense = assets.synthetic_dense_coor()
    up = _maybe_pickle(root, cfg.MISC.UPSAMPLE_PATH)
    if up is None:
        up = assets.synthetic_upsample_weight()
    return (decoder_cls or Decoder)(**(extra or {}), global_feature_dim=encoder_info['global_feature_dim'],
                                    f_in_Dim=encoder_info['fmaps_dim'],
                   f_out_Dim=cfg.MODEL.IMG_DIMS, gcn_in_dim=cfg.MODEL.GCN_IN_DIM, gcn_out_dim=cfg.MODEL.GCN_OUT_DIM,
                   graph_k=cfg.MODEL.graph_k, graph_layer_num=cfg.MODEL.graph_layer_num, vertex_num=778,
                   dense_coor=dense, left_graph_dict=left, right_graph_dict=right, num_attn_heads=4,
                   upsample_weight=torch.from_numpy(up).float(), dropout=cfg.TRAIN.dropout)


def load_model(cfg=None):
    if cfg is None or isinstance(cfg, str):
        cfg = load_cfg(cfg)
    encoder, mid_model = load_encoder(cfg)
    dec = load_decoder(cfg, mid_model.get_info())
    return HandNET_GCN(encoder, mid_model, dec)


def build_model(dropout=0.05, encoder_type='resnet50'):
    """The BASELINE configuration (ResNet50 + attention decoder) without a config file."""
    cfg = load_cfg(None)
    cfg.MODEL.ENCODER_TYPE = encoder_type
    cfg.TRAIN.dropout = dropout
    return load_model(cfg)
