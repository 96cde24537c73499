"""Linear / polynomial warm-up schedule with the interface of the reference's common/lr_scheduler.py
(class name spelled as there)."""


class WarmupSchleduler:
    def __init__(self, optimizer, base_lr, warm_up_iterations, warm_up_polynomial_order):
        self.optimizer = optimizer
        self.base_lr = base_lr
        self.warm_up_iterations = warm_up_iterations
        self.warm_up_polynomial_order = warm_up_polynomial_order
        self._last_lr = None

    @property
    def current_lr(self):
        return self._last_lr

    def update_learning_rate(self, iteration_count):
        warm, order = self.warm_up_iterations, self.warm_up_polynomial_order
        if warm is not None and order is not None and warm > 0 and iteration_count <= warm:
            lr = ((iteration_count / warm) ** order) * self.base_lr
        else:
            lr = self.base_lr
        self._last_lr = lr
        for param_group in self.optimizer.param_groups:
            param_group["lr"] = lr


WarmupScheduler = WarmupSchleduler
