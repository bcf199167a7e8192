"""Adam as used by the runners (lasagne.updates.adam; formula reference custom/updates.py:73-99) lives in
csrc/elementwise.hip::adam_kernel and is applied by ``AdeNetModel.train_step`` / ``apply_adam``.
Per-layer learning rates (``adam_vlr`` / ``generate_lr_map``, reference custom/updates.py:10-99, used only by
runners/1stream_variable_lr.py) are a later-round item."""

BETA1, BETA2, EPSILON = 0.9, 0.999, 1e-8


def adam(model, learning_rate=1e-3):
    """Returns the update callable for ``model`` (one Adam step on its current gradient buffer)."""
    def step():
        model.apply_adam(learning_rate)
    return step
