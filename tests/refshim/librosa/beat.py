def beat_track(*a, **k):
    raise NotImplementedError("beat tracking is host post-processing outside the path")
