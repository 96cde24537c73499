"""Warm-up learning-rate schedule with the public surface of the reference's common/lr_scheduler.py
(`WarmupSchleduler` - spelled as there - with `update_learning_rate(it)` and `current_lr`).

    lr(it) = base_lr * (it / warm) ** order      for 0 <= it <= warm, warm > 0
           = base_lr                              otherwise

The rule lives in one pure function (`warmup_factor`), so it is testable without an optimizer; the class only writes the
result into every parameter group, which is how both torch.optim.Adam and optim.FusedAdam read it."""


def warmup_factor(iteration, warm_up_iterations, order):
    """Fraction of the base rate at `iteration` (1.0 once the warm-up is over or when it is disabled)."""
    disabled = warm_up_iterations is None or order is None or warm_up_iterations <= 0
    if disabled or iteration > warm_up_iterations:
        return 1.0
    return (iteration / warm_up_iterations) ** order


class WarmupSchleduler:
    def __init__(self, optimizer, base_lr, warm_up_iterations, warm_up_polynomial_order):
        self.optimizer = optimizer
        self.base_lr = base_lr
        self.warm_up_iterations = warm_up_iterations
        self.warm_up_polynomial_order = warm_up_polynomial_order
        self.current_lr = None  # set by the first update, like the reference's property before any update

    def update_learning_rate(self, iteration_count):
        self.current_lr = warmup_factor(iteration_count, self.warm_up_iterations, self.warm_up_polynomial_order) * self.base_lr
        for group in self.optimizer.param_groups:
            group["lr"] = self.current_lr
        return self.current_lr


WarmupScheduler = WarmupSchleduler
