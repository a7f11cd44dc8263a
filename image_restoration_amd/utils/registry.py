"""name -> class registries, the plugin seam of the path.

Same observable behaviour as the reference's ``basicsr/utils/registry.py`` (Registry :4-75, the five global
registries :78-82): ``@REG.register()`` or ``REG.register(obj)`` files an object under its ``__name__``, filing a
name twice asserts, looking up an unknown name raises ``KeyError``.
"""


class Registry:
    """A named table of classes / functions keyed by their ``__name__``."""

    def __init__(self, name):
        self.name = name
        self.table = {}

    def _file(self, obj):
        key = obj.__name__
        assert key not in self.table, f"An object named '{key}' was already registered in '{self.name}' registry!"
        self.table[key] = obj
        return obj

    def register(self, obj=None):
        # decorator form returns the filing function itself (it hands the object back); call form files right away
        if obj is None:
            return self._file
        self._file(obj)
        return None

    def get(self, name):
        try:
            return self.table[name]
        except KeyError:
            raise KeyError(f"No object named '{name}' found in '{self.name}' registry!") from None

    def __contains__(self, name):
        return name in self.table

    def __iter__(self):
        return iter(self.table.items())

    def keys(self):
        return self.table.keys()


ARCH_REGISTRY, MODEL_REGISTRY, LOSS_REGISTRY, DATASET_REGISTRY, METRIC_REGISTRY = (
    Registry(kind) for kind in ('arch', 'model', 'loss', 'dataset', 'metric'))


def instantiate(registry, opt, what, type_key='type', as_kwargs=True):
    """Shared body of build_network / build_loss / build_model: look ``opt[type_key]`` up in ``registry`` and construct it —
    from the remaining keys as keyword arguments (architectures, losses) or from the whole option dict (models) — and log
    the reference's "<what> [<class>] is created." line.  ``opt`` is not modified."""
    import copy
    import logging
    conf = copy.deepcopy(opt)
    cls = registry.get(conf.pop(type_key) if as_kwargs else conf[type_key])
    obj = cls(**conf) if as_kwargs else cls(conf)
    logging.getLogger('basicsr').info(f'{what} [{type(obj).__name__}] is created.')
    return obj
