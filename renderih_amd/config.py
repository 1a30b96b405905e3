"""Minimal yacs-compatible config node (yacs is not installed here) + the reference's defaults.

`load_cfg(path)` mirrors utils/config.py:7-21: defaults merged with a user YAML.  Only the keys the model code
reads are defaulted (utils/defaults.yaml MODEL/TRAIN/MISC blocks); any other key in a user file is accepted.
"""
import copy
import yaml


class CfgNode(dict):
    def __init__(self, init=None, new_allowed=True):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def merge(self, other):
        for k, v in other.items():
            if isinstance(v, dict) and isinstance(self.get(k), CfgNode):
                self[k].merge(v)
            else:
                self[k] = CfgNode(v) if isinstance(v, dict) else v

    def merge_from_file(self, path):
        with open(path) as f:
            self.merge(yaml.safe_load(f) or {})

    def set_new_allowed(self, flag):
        pass

    def clone(self):
        return copy.deepcopy(self)

    def dump(self):
        def plain(n):
            return {k: plain(v) if isinstance(v, dict) else v for k, v in n.items()}
        return yaml.safe_dump(plain(self))


_DEFAULTS = {
    'MISC': {'MANO_PATH': 'misc/mano', 'GRAPH_LEFT_DICT_PATH': 'misc/graph_left.pkl',
             'GRAPH_RIGHT_DICT_PATH': 'misc/graph_right.pkl', 'DENSE_COLOR': 'misc/v_color.pkl',
             'UPSAMPLE_PATH': 'misc/upsample.pkl'},
    'MODEL': {'ENCODER_TYPE': 'resnet50', 'DECONV_DIMS': [256, 256, 256, 256], 'IMG_DIMS': [256, 128, 64],
              'GCN_IN_DIM': [512, 256, 128], 'GCN_OUT_DIM': [256, 128, 64], 'ENCODER_PRETRAIN_PATH': 'none',
              'freeze_upsample': True, 'graph_k': 2, 'graph_layer_num': 4},
    'MODEL_PARAM': {'MODEL_PRETRAIN_PATH': 'none'},
    'TRAIN': {'dropout': 0.05, 'BATCH_SIZE': 64, 'LR': 3.0e-4, 'weight_decay': 1.0e-2},
}


def get_cfg_defaults():
    return CfgNode(_DEFAULTS)


def load_cfg(path=None):
    cfg = get_cfg_defaults()
    if path is not None:
        cfg.merge_from_file(path)
    return cfg
