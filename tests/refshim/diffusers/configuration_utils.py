import functools
import inspect
from types import SimpleNamespace


class ConfigMixin:
    pass


def register_to_config(init):
    """Records the constructor arguments as `self.config`, as diffusers' decorator does."""
    @functools.wraps(init)
    def wrapped(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        self.config = SimpleNamespace(**{k: v for k, v in bound.arguments.items() if k != "self"})
        init(self, *args, **kwargs)
    return wrapped
