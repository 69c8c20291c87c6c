"""Name -> factory registries (reference modeling/registry.py:5-12, utils/registry.py:17-45)."""


class Registry(dict):
    def register(self, name, fn=None):
        if fn is not None:
            self[name] = fn
            return fn

        def deco(f):
            self[name] = f
            return f
        return deco


BACKBONES = Registry()
RPN_HEADS = Registry()
ROI_BOX_FEATURE_EXTRACTORS = Registry()
ROI_BOX_PREDICTOR = Registry()
ROI_MASK_FEATURE_EXTRACTORS = Registry()
ROI_MASK_PREDICTOR = Registry()
