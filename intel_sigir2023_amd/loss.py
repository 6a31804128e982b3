"""Loss classes -- host-side mirror of the reference's ``loss/`` package.

``BPRloss`` / ``Listloss`` / ``IntBPRloss`` / ``IntListloss`` keep the reference's contract
(``parse_loss_args``, ``__init__(args)``, ``__call__(out_dict, in_batch) -> (loss, ensemble_loss,
intent_loss)`` with ``loss.backward()`` support) and evaluate in HIP kernels (csrc/loss.hip), forward and
gradient in one launch.  The BPR negative-sampling noise (``torch.rand_like`` in loss/BPRloss.py:26) is
drawn on the device unless ``in_batch['bpr_noise']`` supplies it (parity tests pass the reference's draw).
"""
import torch
import torch.nn as nn

from . import _lib as L


def _ws(device, B, Lmax, K):
    nb = L.lib().intel_loss_workspace_bytes(B, Lmax, K)
    return torch.empty(int(nb), dtype=torch.uint8, device=device), nb


def _i32(t):
    return t if t.dtype == torch.int32 and t.is_contiguous() else t.to(torch.int32).contiguous()


def _scores(in_batch):
    s = in_batch['scores']
    if s.dtype == torch.float64:
        return s.contiguous(), None
    return None, s.float().contiguous()


class _PairLossFn(torch.autograd.Function):
    """ens/weights -> scalar loss; the kernels return the gradient with the forward value."""

    @staticmethod
    def forward(ctx, ens, weights, kind, ranking, session_len, sc64, sc32, noise, cal_div, alpha):
        L.require_gpu(ens)
        B, Lmax = ens.shape
        K = weights.shape[2]
        dev = ens.device
        ens_c, w_c = ens.detach().float().contiguous(), weights.detach().float().contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        d_ens = torch.empty(B, Lmax, dtype=torch.float32, device=dev)
        d_w = torch.empty(B, Lmax, K, dtype=torch.float32, device=dev)
        ws, nb = _ws(dev, B, Lmax, K)
        lib = L.lib()
        if kind == 'bpr':
            select = torch.empty(B, Lmax, dtype=torch.int32, device=dev)
            L.check(lib.intel_bpr_loss(B, Lmax, K, L.ptr(ens_c), L.ptr(ranking), L.ptr(session_len), L.ptr(noise),
                                       L.ptr(sc64), L.ptr(sc32), L.ptr(w_c), int(cal_div), float(alpha), 1.0,
                                       L.ptr(loss), L.ptr(select), L.ptr(d_ens), L.ptr(d_w), L.ptr(ws), nb,
                                       L.stream_ptr(dev)), 'intel_bpr_loss')
            ctx.select = select
        elif kind == 'mse':
            L.check(lib.intel_mse_loss(B, Lmax, K, L.ptr(ens_c), L.ptr(ranking), L.ptr(session_len), L.ptr(sc64),
                                       L.ptr(sc32), L.ptr(w_c), int(cal_div), float(alpha), 1.0, L.ptr(loss),
                                       L.ptr(d_ens), L.ptr(d_w), L.ptr(ws), nb, L.stream_ptr(dev)), 'intel_mse_loss')
        else:
            L.check(lib.intel_list_loss(B, Lmax, K, L.ptr(ens_c), L.ptr(ranking), L.ptr(session_len), L.ptr(sc64),
                                        L.ptr(sc32), L.ptr(w_c), int(cal_div), float(alpha), 1.0, L.ptr(loss),
                                        L.ptr(d_ens), L.ptr(d_w), L.ptr(ws), nb, L.stream_ptr(dev)), 'intel_list_loss')
        ctx.save_for_backward(d_ens, d_w)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        d_ens, d_w = ctx.saved_tensors
        g = g.float()
        return d_ens * g, d_w * g, None, None, None, None, None, None, None, None


class _IntentLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, label, kl_weight, kl_temp):
        L.require_gpu(pred)
        B, I = pred.shape
        dev = pred.device
        out3 = torch.empty(3, dtype=torch.float64, device=dev)
        d_pred = torch.empty(B, I, dtype=torch.float32, device=dev)
        ws, nb = _ws(dev, B, 1, 1)
        pred_c, label_c = pred.detach().float().contiguous(), label.double().contiguous()   # keep alive
        L.check(L.lib().intel_intent_loss(B, I, L.ptr(pred_c),
                                          L.ptr(label_c), float(kl_weight), float(kl_temp), 1.0,
                                          L.ptr(out3), L.ptr(d_pred), L.ptr(ws), nb, L.stream_ptr(dev)), 'intel_intent_loss')
        ctx.save_for_backward(d_pred)
        ctx.mark_non_differentiable(out3)
        return out3[0].clone(), out3

    @staticmethod
    def backward(ctx, g, _g3):
        (d_pred,) = ctx.saved_tensors
        return d_pred * g.float(), None, None, None


class Baseloss(nn.Module):
    """loss/Baseloss.py:6-19."""

    @staticmethod
    def parse_loss_args(parser):
        parser.add_argument('--cal_diversity', type=int, default=0)
        parser.add_argument('--diversity_alpha', type=float, default=0.01)
        return parser

    def __init__(self, args):
        super().__init__()
        self.cal_diversity = args.cal_diversity
        self.diversity_alpha = args.diversity_alpha


class BaseIntloss(Baseloss):
    """loss/BaseIntloss.py:10-70."""

    @staticmethod
    def parse_loss_args(parser):
        parser.add_argument('--intent_weight', type=float, default=0.1, help='Weight for intent loss.')
        parser.add_argument('--ensemble_weight', type=float, default=1, help='Weight for ensemble loss.')
        parser.add_argument('--kl_temp', type=float, default=2)
        parser.add_argument('--kl_weight', type=float, default=0.5)
        return Baseloss.parse_loss_args(parser)

    def __init__(self, args):
        super().__init__(args)
        self.intent_weight = args.intent_weight
        self.ensemble_weight = args.ensemble_weight
        self.kl_weight, self.T = args.kl_weight, args.kl_temp

    def get_intloss(self, out_dict, in_batch):
        loss, out3 = _IntentLossFn.apply(out_dict['intents'], in_batch['intents'], self.kl_weight, self.T)
        return loss, out3[1], out3[2]

    def _pair(self, kind, out_dict, in_batch):
        ens, weights = out_dict['ens_score'], out_dict['weights']
        sc64, sc32 = _scores(in_batch)
        noise = None
        if kind == 'bpr':
            noise = in_batch.get('bpr_noise')
            if noise is None:
                B, Lmax = ens.shape
                noise = torch.rand(B, Lmax, Lmax, dtype=torch.float32, device=ens.device)
            noise = noise.to(ens.device).float().contiguous()
        loss = _PairLossFn.apply(ens, weights, kind, _i32(in_batch['ranking']), _i32(in_batch['session_len']),
                                 sc64, sc32, noise, self.cal_diversity, self.diversity_alpha)
        return loss


class BPRloss(BaseIntloss):
    """loss/BPRloss.py:7-56."""

    def forward(self, out_dict, in_batch):
        loss = self._pair('bpr', out_dict, in_batch)
        return loss, loss, loss


class Listloss(BaseIntloss):
    """loss/Listloss.py:7-43."""

    def forward(self, out_dict, in_batch):
        loss = self._pair('list', out_dict, in_batch)
        return loss, loss, loss


class MSEloss(BaseIntloss):
    """loss/MSEloss.py:7-30."""

    def forward(self, out_dict, in_batch):
        loss = self._pair('mse', out_dict, in_batch)
        return loss, loss, loss


class IntMSEloss(MSEloss):
    """loss/IntMSEloss.py:11-21."""

    def forward(self, out_dict, in_batch):
        intent_loss, _, _ = self.get_intloss(out_dict, in_batch)
        ensemble_loss = self._pair('mse', out_dict, in_batch)
        loss = ensemble_loss * self.ensemble_weight + intent_loss * self.intent_weight
        return loss, ensemble_loss, intent_loss


class IntBPRloss(BPRloss):
    """loss/IntBPRloss.py:10-20."""

    def forward(self, out_dict, in_batch):
        intent_loss, _, _ = self.get_intloss(out_dict, in_batch)
        ensemble_loss = self._pair('bpr', out_dict, in_batch)
        loss = ensemble_loss * self.ensemble_weight + intent_loss * self.intent_weight
        return loss, ensemble_loss, intent_loss


class IntListloss(Listloss):
    """loss/IntListloss.py:9-19."""

    def forward(self, out_dict, in_batch):
        intent_loss, _, _ = self.get_intloss(out_dict, in_batch)
        ensemble_loss = self._pair('list', out_dict, in_batch)
        loss = ensemble_loss * self.ensemble_weight + intent_loss * self.intent_weight
        return loss, ensemble_loss, intent_loss


def bpr_select_index(ens, in_batch, noise):
    """The sampled negative per row (loss/BPRloss.py:26-28) -- exposed for tests."""
    B, Lmax = ens.shape
    dev = ens.device
    K = in_batch['scores'].shape[2]
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    select = torch.empty(B, Lmax, dtype=torch.int32, device=dev)
    ws, nb = _ws(dev, B, Lmax, K)
    # temporaries must stay referenced until the launch is enqueued (a freed block can be re-used
    # by the next conversion before the kernel has read it)
    ens_c, rank_c, len_c, noise_c = ens.float().contiguous(), _i32(in_batch['ranking']), _i32(in_batch['session_len']), noise.float().contiguous()
    L.check(L.lib().intel_bpr_loss(B, Lmax, K, L.ptr(ens_c), L.ptr(rank_c),
                                   L.ptr(len_c), L.ptr(noise_c), None, None,
                                   None, 0, 0.0, 1.0, L.ptr(loss), L.ptr(select), None, None, L.ptr(ws), nb,
                                   L.stream_ptr(dev)), 'intel_bpr_loss')
    return select
