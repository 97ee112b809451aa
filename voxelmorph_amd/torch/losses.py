"""Drop-in mirror of `voxelmorph/torch/losses.py` (reference) on the MI355X HIP kernels.

Every class keeps the reference's `loss(y_true, y_pred) -> 0-dim tensor` contract.
"""
from . import functional as VF
from . import planar as VP


class NCC:
    """Local (over window) normalized cross correlation loss (reference: losses.py:7-67)."""

    def __init__(self, win=None):
        self.win = win

    def loss(self, y_true, y_pred):
        ndims = len(list(y_true.size())) - 2
        assert ndims in [1, 2, 3], "volumes should be 1 to 3 dimensions. found: %d" % ndims
        win = [9] * ndims if self.win is None else list(self.win)
        if len(win) != ndims or len(set(win)) != 1 or int(win[0]) % 2 == 0 or int(win[0]) != win[0]:
            # the reference pads every axis by win[0] // 2 whatever the other window sizes are (losses.py:31-36): windows that are not
            # odd and of one size change the SHAPE of the box sums -- the general separable passes follow that rule
            if len(win) != ndims:
                # (torch.ones([1, 1, *win]) of another rank makes the reference's conv raise)
                raise ValueError("NCC: %d window sizes for a %d-D volume" % (len(win), ndims))
            return VF.NCCWinFn.apply(y_true, y_pred, [int(w) for w in win])
        if ndims == 1:
            return VP.NCC1dFn.apply(y_true, y_pred, int(win[0]))
        if ndims == 2:
            return VP.NCC2dFn.apply(y_true, y_pred, int(win[0]))
        return VF.NCCFn.apply(y_true, y_pred, int(win[0]))


class MSE:
    """Mean squared error loss (reference: losses.py:70-76)."""

    def loss(self, y_true, y_pred):
        return VF.MSEFn.apply(y_true, y_pred)


class Dice:
    """N-D dice for segmentation (reference: losses.py:79-90)."""

    def loss(self, y_true, y_pred):
        return VF.DiceFn.apply(y_true, y_pred)


class Grad:
    """N-D gradient loss (reference: losses.py:93-135)."""

    def __init__(self, penalty='l1', loss_mult=None):
        self.penalty = penalty
        self.loss_mult = loss_mult

    def loss(self, _, y_pred):
        if self.penalty != 'l1':
            assert self.penalty == 'l2', 'penalty can only be l1 or l2. Got: %s' % self.penalty
        mult = 1.0 if self.loss_mult is None else float(self.loss_mult)
        if y_pred.dim() == 4:
            return VP.GradLoss2dFn.apply(y_pred, self.penalty, mult)
        return VF.GradLossFn.apply(y_pred, self.penalty, mult)


def weighted_sum(terms, weights, running=None):
    """The reference loop's `loss += loss_function(y_true[n], y_pred[n]) * weights[n]` (scripts/torch/train.py:205-212) for terms already
    evaluated: one HIP launch forward, one backward, no ATen kernel (`a + w * b` on 0-dim tensors costs a mul and an add launch each way).
    `running`: optional device tensor of len(terms) + 1 floats that accumulates the weighted terms and the total (per-epoch logging)."""
    return VF.WeightedSumFn.apply(tuple(float(w) for w in weights), running, *terms)
