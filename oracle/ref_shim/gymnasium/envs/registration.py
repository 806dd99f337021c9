import importlib

registry = {}


class EnvSpec:
    def __init__(self, id, entry_point, kwargs):
        self.id = id
        self.entry_point = entry_point
        self.kwargs = dict(kwargs or {})


def register(id, entry_point=None, kwargs=None, **_ignored):
    registry[id] = EnvSpec(id, entry_point, kwargs)


def make(id, **kwargs):
    spec = registry[id]
    ep = spec.entry_point
    if isinstance(ep, str):
        mod_name, attr = ep.split(":")
        ep = getattr(importlib.import_module(mod_name), attr)
    kw = dict(spec.kwargs)
    kw.update(kwargs)
    env = ep(**kw)
    env.spec = spec
    return env
