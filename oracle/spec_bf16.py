"""TEST INFRASTRUCTURE (checker, never shipped): the arithmetic of the bf16-MFMA MLP mode (NarutoFieldDesc.mlp_mode = NARUTO_MLP_BF16),
restated in torch next to the exact-mode oracle (oracle/spec_torch.py), with a bound that pins it to the exact network.

What the speed mode changes -- and nothing else -- is where operands are rounded to bf16 (round to nearest even, ``v_cvt_pk_bf16_f32``)
before a matrix product with fp32 accumulation (``v_mfma_f32_32x32x16_bf16``).  Rounding points in the kernels
(naruto_amd/csrc/naruto_field.hip):

  forward  k_query_fwd_bf / fwd_tile_bf
    * the weights of sdf layer 0, sdf layer 1 and colour layer 0, when staged into LDS (stage_fwd_weights_bf: ``pk_bf16`` of eight
      consecutive K entries)                                                                         -> bf(W) below
    * the 32 hash features and the 48 OneBlob values entering sdf layer 0 / colour layer 0 (``pack8(fa)``, ``pack8(lo8)`` ...)   -> bf(x)
    * relu(h) entering sdf layer 1 (``pack8_acc<true>(hA, 8 * kb)``)                                 -> bf(relu(h))
    * the 15 geo outputs entering colour layer 0 (``pack8_acc<false>(oA, 0)``)                        -> bf(out[:, 1:])
    * NOT rounded: the encodings themselves, the sdf / geo OUTPUTS (fp32 accumulators), the 32 -> 3 colour layer (fp32 VALU code), the
      uncertainty-grid sample, compositing, losses, optimiser.
  backward  k_query_bwd_bf / bwd_tile_bf
    * every cotangent entering a matrix product (``pack8_acc<false>(dcv, ..)``, ``pack8(gq)``, ``dhP``, ``dovP``), the saved layer inputs
      as the other operand of the weight gradients (``pack8_acc<false>(xU, ..)``), the transposed bf16 weight images -> BfLinear.backward;
    * ReLU masks come from the bf16 FORWARD's pre-activations; the colour layer 1's input gradient and weight gradient are fp32.

``BfLinear`` is one such layer; ``query_color_sdf_bf16`` chains them exactly as the kernel does.  ``bf16_forward_bound`` propagates the
half-ulp rounding error 2^-9 of every rounded operand through the network in absolute values: an elementwise bound on
|restatement - exact network| that tests/test_oracle.py checks against an fp64 evaluation of the exact network (so the restatement is
pinned to the reference's decoder, decoder.py:29-41,99-116, not only to the kernel it describes)."""
from __future__ import annotations

import torch

from . import spec_torch as S

BF16_HALF_ULP = 2.0 ** -9          # relative rounding error of one operand (8 significant bits, round to nearest even)


def bf(t: torch.Tensor) -> torch.Tensor:
    """fp32 -> bf16 (nearest even) -> fp32."""
    return t.bfloat16().float()


class BfLinear(torch.autograd.Function):
    """y = bf(x) . bf(W)^T with fp32-or-better accumulation; backward as k_query_bwd_bf computes it: dx = bf(g) . bf(W),
    dW = bf(g)^T . bf(x).  ``exact``: the 32 -> 3 colour layer, whose forward and input gradient are fp32 VALU code (its weight
    gradient still multiplies bf16-rounded cotangents with fp32 inputs on the VALU: kept fp32 here as in the kernel)."""

    @staticmethod
    def forward(ctx, x, W, exact):
        ctx.save_for_backward(x, W)
        ctx.exact = exact
        return (x.double() @ W.double().T).float() if exact else (bf(x).double() @ bf(W).double().T).float()

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        dx = (g.double() @ W.double()).float() if ctx.exact else (bf(g).double() @ bf(W).double()).float()
        dW = (bf(g).double().T @ bf(x).double()).float()
        return dx, dW, None


def query_color_sdf_bf16(ora: "S.OracleField", x: torch.Tensor):
    """raw [M,5] = (rgb pre-sigmoid, sdf, uncertainty sample) and the sdf net's 16 outputs [M,16] in the bf16 mode, differentiable
    w.r.t. the oracle's parameters (the table through the exact hash encode: the gathers and the blend stay fp32)."""
    feats, pos = S.hash_encode(x, ora.table, ora.meta), S.oneblob_encode(x, ora.n_bins)
    h = BfLinear.apply(torch.cat([feats, pos], -1), ora.sdf_w0, False)
    out = BfLinear.apply(torch.relu(h), ora.sdf_w1, False)
    c = BfLinear.apply(torch.cat([pos, out[:, 1:]], -1), ora.col_w0, False)
    rgb = BfLinear.apply(torch.relu(c), ora.col_w1, True)
    unc = S.sample_uncert_grid_ref(ora.uncert_grid, x)[:, None]
    return torch.cat([rgb, out[:, :1], unc], -1), out


def bf16_forward_bound(ora: "S.OracleField", x: torch.Tensor):
    """Elementwise bound on |query_color_sdf_bf16 - exact network| for (rgb[3], sdf): every rounded operand a carries |bf(a) - a| <=
    2^-9 |a|, so a product of two rounded operands is off by at most (2 * 2^-9 + 2^-18) |a| |w|; errors of a layer's inputs pass through
    |W| (ReLU is 1-Lipschitz).  fp32 accumulation noise (~1e-7 relative) rides on top: callers add it as slack.  Returns (bound [M,4],
    exact [M,4]) evaluated in float64."""
    e = 2.0 * BF16_HALF_ULP + BF16_HALF_ULP ** 2
    with torch.no_grad():
        feats, pos = S.hash_encode(x, ora.table, ora.meta).double(), S.oneblob_encode(x, ora.n_bins).double()
        W0, W1, C0, C1 = (w.detach().double() for w in (ora.sdf_w0, ora.sdf_w1, ora.col_w0, ora.col_w1))
        in0 = torch.cat([feats, pos], -1)
        h = in0 @ W0.T
        dh = e * (in0.abs() @ W0.abs().T)
        r = torch.relu(h)
        out = r @ W1.T
        # the kernel rounds relu(h~) where h~ carries dh: |bf(relu(h~)) - relu(h)| <= dh + 2^-9 (|relu h| + dh)
        dr = dh + BF16_HALF_ULP * (r + dh)
        dout = dr @ W1.abs().T + BF16_HALF_ULP * ((r + dr) @ W1.abs().T) * (1.0 + BF16_HALF_ULP)
        in1 = torch.cat([pos, out[:, 1:]], -1)
        din1 = torch.cat([torch.zeros_like(pos), dout[:, 1:]], -1)
        c = in1 @ C0.T
        dc = din1 @ C0.abs().T + e * ((in1.abs() + din1) @ C0.abs().T)
        rgb = torch.relu(c) @ C1.T
        drgb = dc @ C1.abs().T                                   # colour layer 1 is exact fp32
        return torch.cat([drgb, dout[:, :1]], -1), torch.cat([rgb, out[:, :1]], -1)
