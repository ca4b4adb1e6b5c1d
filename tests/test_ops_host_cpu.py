"""Host-side logic of ops.py that needs no GPU (nothing here launches a kernel)."""


def test_zero_grad_drops_weight_gradient_groups_left_by_an_aborted_backward():
    """a backward pass that dies between WgradGroup.add() and flush() leaves operands queued; the next zero_grad must DROP
    them (never flush a stale gradient into the fresh bucket), clear the `_ffvc_deferred` marks a gradient-ready listener obeys, and
    release the operands.  Host logic only: nothing is launched."""
    import types

    import torch

    from feed_forward_vqgan_clip_amd import ops

    def pack():
        return types.SimpleNamespace(weight=torch.nn.Parameter(torch.zeros(4, 4)), bias=torch.nn.Parameter(torch.zeros(4)), on_grad=None)

    members = [pack(), pack(), pack()]
    g = ops.WgradGroup(members)
    # what add() leaves behind for two of three members when the pass aborts (no CUDA stream is touched here)
    for i in (0, 1):
        g.pending[i] = (torch.zeros(8, 4), torch.zeros(8, 4), 8, False)
        members[i].weight._ffvc_deferred = True
        members[i].bias._ffvc_deferred = True
    ops._PENDING_GROUPS.append(g)
    ops._SIDE["cb"] = True
    try:
        assert ops.discard_wgrad_groups() == 2
        assert g.pending == {} and ops._PENDING_GROUPS == [] and ops._SIDE["cb"] is False
        assert not any(getattr(m.weight, "_ffvc_deferred", False) or getattr(m.bias, "_ffvc_deferred", False) for m in members)
        assert ops.discard_wgrad_groups() == 0
    finally:
        del ops._PENDING_GROUPS[:]
        ops._SIDE["cb"] = False
