"""Autograd bridge for ODEBlock.forward.

Inference (no_grad, or nothing requires grad) calls the fused forward and never
materialises the trajectory.  Training saves `sol` as torchdiffeq's
odeint_adjoint does and runs the adjoint-Euler backward in
libgnode_hip.so.
"""
from __future__ import annotations

import torch

from . import ops


def _needs_grad(params: dict) -> bool:
    return torch.is_grad_enabled() and any(p.requires_grad for p in params.values())


class _GNODEForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, graph, x2d, dts, method, out_rows, keys, *tensors):
        params = dict(zip(keys, tensors))
        S, I, R, sol = ops.forward(graph, x2d, params, dts, method, out_rows, want_sol=True)
        ctx.graph, ctx.dts, ctx.method, ctx.out_rows, ctx.keys = graph, dts, method, out_rows, keys
        ctx.keep = sol.gnode_keep            # kept activations (a plain buffer nothing else references), or None
        ctx.save_for_backward(x2d, sol, *tensors)
        return S, I, R

    @staticmethod
    def backward(ctx, gS, gI, gR):
        x2d, sol, *tensors = ctx.saved_tensors
        params = dict(zip(ctx.keys, tensors))
        ref = next(g for g in (gS, gI, gR) if g is not None)          # an output the loss did not use has no gradient
        gS, gI, gR = (torch.zeros_like(ref) if g is None else g for g in (gS, gI, gR))
        grads = ops.backward(ctx.graph, x2d, params, ctx.dts, ctx.method, ctx.out_rows, sol,
                             gS.contiguous(), gI.contiguous(), gR.contiguous(), keep=ctx.keep)
        # (ctx.keep stays: a second backward through this node -- retain_graph=True, two losses on one forward --
        #  needs it again; it is freed with ctx and sol)
        return (None, None, None, None, None, None, *[grads[k] for k in ctx.keys])


def forward_with_grad(graph, x2d, params, dts, method="euler", out_rows=None):
    if not _needs_grad(params):
        with torch.no_grad():
            S, I, R, _ = ops.forward(graph, x2d, {k: v.detach() for k, v in params.items()}, dts, method, out_rows)
        return S, I, R
    keys = tuple(params.keys())
    return _GNODEForward.apply(graph, x2d, dts, method, out_rows, keys, *[params[k] for k in keys])


class _L1LossSum(torch.autograd.Function):
    """sum |cat(S, I, R)[rows, T, 3][:, t0:, :] - y[:, t0:, :]| with its gradient from one kernel (csrc/gnode_loss.hip)."""

    @staticmethod
    def forward(ctx, S, I, R, y, t0):
        need_grad = any(ctx.needs_input_grad[:3])
        total, sgn = ops.l1_loss_sum(S, I, R, y, t0, want_sign=need_grad)
        ctx.shape = S.shape
        if need_grad:
            ctx.save_for_backward(sgn)
        return total

    @staticmethod
    def backward(ctx, g):
        (sgn,) = ctx.saved_tensors
        gs = (sgn * g.to(torch.float32)).view(3, *ctx.shape)
        return gs[0], gs[1], gs[2], None, None


def l1_loss_sum(S, I, R, y, t0=1):
    """The reference's loss numerator (ode_nn_ngraph_sim.py:230-234): S, I, R are the model's [T, rows(, 1)] outputs,
    y the labels [rows, T, 3] (fp32 or fp64); returns a float64 scalar.  Divide by rows * (T - t0) * 3 for L1Loss's mean."""
    return _L1LossSum.apply(S, I, R, y, t0)


def l1_loss_mean_backward(S, I, R, y, count, t0=1):
    """loss.backward() for  loss = l1_loss_sum(S, I, R, y, t0) / count  (the trainer's step, ode_nn_ngraph_sim.py:234-236) without
    the scalar graph in between: the loss kernel writes sign(pred - y) / count, which IS dloss/d(S, I, R), and the backward
    of the model starts from it -- five tiny launches less per step (division, ones, division's backward, a cast, a multiply).
    Returns the loss SUM (float64 device scalar), as l1_loss_sum does.  Bit-identical gradients: the scalar path multiplies
    the signs by the same fp32(1 / count)."""
    total, sgn = ops.l1_loss_sum(S, I, R, y, t0, want_sign=True, sign_scale=1.0 / float(count))
    gs = sgn.view(3, *S.shape)
    torch.autograd.backward([S, I, R], [gs[0], gs[1], gs[2]])
    return total
