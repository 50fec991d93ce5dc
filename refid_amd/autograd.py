"""Autograd hand-off of the HIP engines, shared by the drop-in architecture modules.

The engine computes every parameter gradient itself (hand-written BPTT into a flat arena).  To the rest of
PyTorch the network must still look like an ordinary differentiable op: ``loss.backward()`` has to deliver the
gradients THROUGH autograd -- one gradient per nn.Parameter, accumulated into ``p.grad`` by autograd's own
AccumulateGrad nodes -- because that is what everything the reference stacks on top relies on:
``torch.nn.parallel.DistributedDataParallel`` (base_model.py:66-72 wraps ``net_g`` when ``opt['dist']``; its reducer
hangs on the AccumulateGrad post-hooks), gradient accumulation over several ``backward()`` calls,
``clip_grad_norm_`` and the torch optimizers (twoImage_event_recurrent_model.py:303-309).

So the Function takes all parameters as inputs and returns their gradients: views of ONE private copy of the flat
gradient arena made right after BPTT (autograd may adopt a returned tensor as ``p.grad``; a copy keeps the arena --
which the next backward overwrites -- from ever being aliased by ``p.grad``).
"""
import torch


class HipNetFunction(torch.autograd.Function):
    """forward(x, event, net, *params) -> net.engine.forward(x, event); backward -> (None, None, None, *dparams)."""

    @staticmethod
    def forward(ctx, x, event, net, *params):
        ctx.net = net
        return net._engine.forward(x, event, save=True)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        net = ctx.net
        eng = net.engine
        eng.zero_grad()                                    # the arena holds THIS backward only; autograd accumulates
        eng.backward(gout, grad_sync=net._grad_sync)
        flat = eng.arena.flat_g.clone()
        grads = []
        for (k, p), need in zip(net._params.items(), ctx.needs_input_grad[3:]):
            if need:
                o, n = eng.arena.offsets[k]
                grads.append(flat[o:o + n].view(eng.arena.shapes[k]))
            else:
                grads.append(None)
        return (None, None, None, *grads)


def apply(net, x, event):
    return HipNetFunction.apply(x, event, net, *net._params.values())
