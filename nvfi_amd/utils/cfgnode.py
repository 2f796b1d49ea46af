"""Minimal attribute-access config node (the reference uses a YACS-style CfgNode, utils/cfgnode.py:115-119)."""


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    @staticmethod
    def load_yaml(path):
        import yaml
        with open(path) as f:
            return CfgNode(yaml.safe_load(f))
