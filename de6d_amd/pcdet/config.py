"""YAML -> attribute-dict configuration, API of core/pcdet/config.py:7-85
(`cfg`, `cfg_from_yaml_file`, `cfg_from_list`, `merge_new_config`, `log_config_to_file`).

`easydict` is not a dependency here: `EasyDict` below is a small attribute dict with the same
recursive-wrapping behaviour the reference relies on (`cfg.MODEL.BACKBONE_3D.SA_CONFIG.get(...)`).
"""
import ast
from pathlib import Path

import yaml


class EasyDict(dict):
    """dict whose keys are also attributes; nested dicts (also inside lists/tuples) are wrapped."""

    def __init__(self, d=None, **kwargs):
        super().__init__()
        src = dict(d or {})
        src.update(kwargs)
        for k, v in src.items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __delattr__(self, k):
        del self[k]

    def update(self, other=None, **kwargs):
        for k, v in dict(other or {}, **kwargs).items():
            self[k] = v

    def setdefault(self, k, default=None):
        if k not in self:
            self[k] = default
        return self[k]


def log_config_to_file(cfg, pre='cfg', logger=None):
    for key in cfg:
        val = cfg[key]
        if isinstance(val, EasyDict):
            logger.info('\n%s.%s = edict()' % (pre, key))
            log_config_to_file(val, pre='%s.%s' % (pre, key), logger=logger)
        else:
            logger.info('%s.%s: %s' % (pre, key, val))


def cfg_from_list(cfg_list, config):
    """`--set KEY VAL KEY VAL ...` overrides with the reference's typing rules (config.py:16-48)."""
    assert len(cfg_list) % 2 == 0
    for dotted, raw in zip(cfg_list[0::2], cfg_list[1::2]):
        *parents, leaf = dotted.split('.')
        node = config
        for name in parents:
            assert name in node, 'NotFoundKey: %s' % name
            node = node[name]
        assert leaf in node, 'NotFoundKey: %s' % leaf
        try:
            value = ast.literal_eval(raw)
        except Exception:
            value = raw
        current = node[leaf]
        if type(value) != type(current) and isinstance(current, EasyDict):
            for pair in str(raw).split(','):
                sub_key, sub_val = pair.split(':')
                current[sub_key] = type(current[sub_key])(sub_val)
        elif type(value) != type(current) and isinstance(current, list):
            node[leaf] = [type(current[0])(x) for x in str(raw).split(',')]
        else:
            assert type(value) == type(current), \
                'type {} does not match original type {}'.format(type(value), type(current))
            node[leaf] = value


def _read_yaml(path):
    with open(path, 'r') as f:
        return yaml.safe_load(f)


def merge_new_config(config, new_config):
    if '_BASE_CONFIG_' in new_config:
        config.update(EasyDict(_read_yaml(new_config['_BASE_CONFIG_'])))
    for key, val in new_config.items():
        if isinstance(val, dict):
            if key not in config:
                config[key] = EasyDict()
            merge_new_config(config[key], val)
        else:
            config[key] = val
    return config


def cfg_from_yaml_file(cfg_file, config):
    merge_new_config(config=config, new_config=_read_yaml(cfg_file))
    return config


cfg = EasyDict()
cfg.ROOT_DIR = (Path(__file__).resolve().parent / '../').resolve()
cfg.LOCAL_RANK = 0
