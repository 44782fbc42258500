from collections import OrderedDict


class BaseOutput(OrderedDict):
    """diffusers' output container: a dict whose keys are also attributes."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)
