"""Early-stopping predicates (reference utils/regularization.py:1-23)."""


def early_stop(cost_window):
    """True when the window (>= 2 entries) is strictly increasing."""
    costs = list(cost_window[i] for i in range(len(cost_window)))
    if len(costs) < 2:
        return False
    return all(a < b for a, b in zip(costs, costs[1:]))


def early_stop2(cost_window, min_val_cost, threshold):
    """True once ``threshold`` entries of the window exceed the best validation cost.
    Like the reference, falls through (returns None, falsy) when the count is never reached
    and returns False for windows shorter than 2."""
    n = len(cost_window)
    if n < 2:
        return False
    worse = 0
    for i in range(n):
        if cost_window[i] > min_val_cost:
            worse += 1
        if worse == threshold:
            return True
    return None
