"""temporal_softmax_loss (reference custom/objectives.py:4-39) is fused into the HIP softmax/loss kernel
(csrc/elementwise.hip); this module exposes it with the reference's name for callers that hold a model."""


def temporal_softmax_loss(model, inputs, targets, mask, window):
    """Masked per-frame cross-entropy of the (already soft-maxed, then soft-maxed again) predictions,
    normalised by the number of valid frames."""
    return model.loss(inputs, targets, mask, window)
