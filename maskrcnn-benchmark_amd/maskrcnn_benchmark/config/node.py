"""A small self-contained configuration tree with the subset of the yacs `CfgNode` behaviour the
reference relies on (yacs is not installed here): attribute access, `merge_from_file` (yaml),
`merge_from_list` (KEY VALUE pairs from the command line, reference tools/train_net.py:165-166),
`freeze`/`defrost`, `clone`, `dump`.  Unknown keys are rejected, like yacs does, so that a typo in
a yaml file fails loudly."""
import ast
import copy

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super(CfgNode, self).__init__()
        object.__setattr__(self, "_frozen", False)
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    # -- attribute protocol
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        if self._frozen:
            raise AttributeError("attempted to set %s on a frozen CfgNode" % name)
        self[name] = value

    # -- life cycle
    def freeze(self):
        self._set_frozen(True)

    def defrost(self):
        self._set_frozen(False)

    def is_frozen(self):
        return self._frozen

    def _set_frozen(self, flag):
        object.__setattr__(self, "_frozen", flag)
        for v in self.values():
            if isinstance(v, CfgNode):
                v._set_frozen(flag)

    def clone(self):
        return copy.deepcopy(self)

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            dict.__setitem__(out, k, copy.deepcopy(v, memo))
        object.__setattr__(out, "_frozen", self._frozen)
        return out

    # -- merging
    @staticmethod
    def _coerce(new, old, key):
        """Bring `new` to the type of the default `old` (tuple<->list, int->float); reject others."""
        if old is None or type(new) is type(old):
            return new
        if isinstance(old, tuple) and isinstance(new, list):
            return tuple(new)
        if isinstance(old, list) and isinstance(new, tuple):
            return list(new)
        if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
            return float(new)
        if isinstance(old, (tuple, list)) and isinstance(new, str):
            return CfgNode._coerce(CfgNode._literal(new), old, key)
        raise ValueError("config key %s: cannot merge %r (%s) over default %r (%s)"
                         % (key, new, type(new).__name__, old, type(old).__name__))

    @staticmethod
    def _literal(s):
        if not isinstance(s, str):
            return s
        try:
            return ast.literal_eval(s)
        except (ValueError, SyntaxError):
            return s

    def _merge(self, other, path):
        for k, v in other.items():
            full = ".".join(path + [k])
            if k not in self:
                raise KeyError("Non-existent config key: %s" % full)
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError("config key %s expects a mapping" % full)
                self[k]._merge(v, path + [k])
            else:
                self[k] = self._coerce(self._literal(v) if isinstance(v, str) and not isinstance(self[k], str) else v,
                                       self[k], full)

    def merge_from_other_cfg(self, other):
        self._merge(other, [])

    def merge_from_file(self, filename):
        with open(filename, "r") as f:
            loaded = yaml.safe_load(f) or {}
        self._merge(loaded, [])

    def merge_from_list(self, opts):
        if len(opts) % 2 != 0:
            raise ValueError("merge_from_list expects KEY VALUE pairs, got %r" % (opts,))
        for key, value in zip(opts[0::2], opts[1::2]):
            node = self
            parts = key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError("Non-existent config key: %s" % key)
                node = node[p]
            leaf = parts[-1]
            if leaf not in node:
                raise KeyError("Non-existent config key: %s" % key)
            old = node[leaf]
            new = value if isinstance(old, str) else self._literal(value)
            node[leaf] = self._coerce(new, old, key)

    # -- output
    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else v) for k, v in self.items()}

    def dump(self):
        def plain(x):
            if isinstance(x, dict):
                return {k: plain(v) for k, v in x.items()}
            if isinstance(x, tuple):
                return [plain(i) for i in x]
            return x
        return yaml.safe_dump(plain(self.to_dict()), default_flow_style=None)

    def __repr__(self):
        return "CfgNode(%s)" % dict.__repr__(self)
