class SchedulerMixin:
    pass
