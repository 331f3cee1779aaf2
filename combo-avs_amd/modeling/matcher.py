"""Hungarian matcher, mirroring models/modeling/matcher.py.

MI355X notes: the reference copies one cost matrix to the host and runs SciPy per frame per decoder output
(10 x #GT-frames syncs per step).  Here `match_layers` builds the cost matrices of ALL frames of ALL decoder
outputs on the device, makes ONE device->host copy and solves the tiny LSAPs on the host.
`point_source(n, p)` supplies the uniform random points ([n,p,2] on the target device) so tests can inject the
reference's coordinate stream.
"""
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment
from torch import nn

from ..ops.points import point_sample


def default_point_source(device):
    def src(n, p):
        return torch.rand(n, p, 2, device=device)
    return src


def batch_cost(pred_logits_b, pred_masks_b, labels, gt_masks, point_coords, w_class, w_mask, w_dice):
    """Cost matrix [Q,G] of one frame (matcher.py:93-131); fp32 (autocast disabled in the reference, :121-123)."""
    prob = pred_logits_b.float().softmax(-1)
    cost_class = -prob[:, labels]
    G = gt_masks.shape[0]
    t = point_sample(gt_masks[:, None].float(), point_coords.expand(G, -1, -1)).squeeze(1)
    o = point_sample(pred_masks_b[:, None].float(), point_coords.expand(pred_masks_b.shape[0], -1, -1)).squeeze(1)
    hw = o.shape[1]
    pos, neg = F.softplus(-o), F.softplus(o)
    cost_mask = (pos @ t.T + neg @ (1 - t).T) / hw
    s = o.sigmoid()
    cost_dice = 1 - (2 * (s @ t.T) + 1) / (s.sum(-1)[:, None] + t.sum(-1)[None, :] + 1)
    return w_mask * cost_mask + w_class * cost_class + w_dice * cost_dice


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_mask: float = 1, cost_dice: float = 1, num_points: int = 0):
        super().__init__()
        self.cost_class, self.cost_mask, self.cost_dice = cost_class, cost_mask, cost_dice
        assert cost_class != 0 or cost_mask != 0 or cost_dice != 0, "all costs cant be 0"
        self.num_points = num_points
        self._lsap_status = None  # device word the LSAP kernel ORs its condition bits into (see check_status)

    @torch.no_grad()
    def cost_matrices(self, outputs, targets, point_source=None):
        ps = point_source or default_point_source(outputs["pred_logits"].device)
        costs = []
        for b in range(outputs["pred_logits"].shape[0]):
            pc = ps(1, self.num_points)  # ONE set of points shared by all masks of the frame (matcher.py:107)
            costs.append(batch_cost(outputs["pred_logits"][b], outputs["pred_masks"][b], targets[b]["labels"],
                                    targets[b]["masks"], pc, self.cost_class, self.cost_mask, self.cost_dice))
        return costs

    @torch.no_grad()
    def batched_cost(self, logits, masks, labels, gt, points, mask_base=None):
        """Cost tensors of N = (outputs x frames) assignment problems at once, one fused HIP launch
        (csrc/matcher.hip).  logits [N,Q,K+1], masks [N,Q,h,w], labels [N,Gmax] (padded), gt [N,Gmax,H,W] (zero
        padded), points [N,P,2] -> [N,Q,Gmax] (columns beyond a frame's real G are sliced away by the caller)."""
        from .. import _lib
        # mask_base [N] int64: problem n's maps start at masks.view(-1,h,w)[mask_base[n]] (no gathered copy of the GT frames)
        logits, masks, gt, points = (t.contiguous().float() for t in (logits, masks, gt, points))
        labels = labels.contiguous()
        _lib.require_cuda(logits, masks, labels, gt, points)
        N, Q, K1 = logits.shape
        G, (h, w), (H, W), P = gt.shape[1], masks.shape[-2:], gt.shape[-2:], points.shape[1]
        cost = torch.empty(N, Q, G, device=logits.device, dtype=torch.float32)
        GC = 8  # targets per launch (csrc/matcher.hip keeps 3*G+1 running sums in registers); columns are independent
        for g0 in range(0, G, GC):
            g1 = min(G, g0 + GC)
            gsub = gt[:, g0:g1].contiguous() if (g0, g1) != (0, G) else gt
            lsub = labels[:, g0:g1].contiguous() if (g0, g1) != (0, G) else labels
            csub = cost if (g0, g1) == (0, G) else torch.empty(N, Q, g1 - g0, device=logits.device, dtype=torch.float32)
            t_ws = torch.empty(N, g1 - g0, P, device=logits.device, dtype=torch.float32)
            _lib.check(_lib.lib().combo_matcher_cost_f32(
                logits.data_ptr(), masks.data_ptr(), _lib.ptr(mask_base), lsub.data_ptr(), gsub.data_ptr(), points.data_ptr(), N, Q, K1,
                g1 - g0, h, w,
                H, W, P, self.cost_class, self.cost_mask, self.cost_dice, t_ws.data_ptr(), csub.data_ptr(), _lib.current_stream()),
                "combo_matcher_cost_f32")
            if csub is not cost:
                cost[:, :, g0:g1] = csub
        return cost

    LSAP_DEVICE_MAX_G = 6

    @torch.no_grad()
    def solve_device(self, cost, gcount):
        """Exact assignment on the device (csrc/lsap.hip), no host sync.  cost [N,Q,Gpad] fp32, gcount [N] int32
        -> row_for_col [N,Gpad] int64 (query matched to target g; -1 for padded targets)."""
        from .. import _lib
        cost = cost.contiguous().float()
        _lib.require_cuda(cost, gcount)
        N, Q, Gpad = cost.shape
        out = torch.empty(N, Gpad, device=cost.device, dtype=torch.int64)
        if self._lsap_status is None or self._lsap_status.device != cost.device:
            self._lsap_status = torch.zeros(1, device=cost.device, dtype=torch.int32)
        _lib.check(_lib.lib().combo_lsap_small_f32(cost.data_ptr(), gcount.data_ptr(), N, Q, Gpad, out.data_ptr(),
                                                   self._lsap_status.data_ptr(), _lib.current_stream()), "combo_lsap_small_f32")
        return out

    def check_status(self):
        """Raises what scipy.optimize.linear_sum_assignment raises at matcher.py:133 when a cost matrix was not finite (a
        diverged step), or when a frame had more targets than the device solver takes.  The device solver only sets bits in a
        status word (no sync on the hot path); call this where a host sync is acceptable - trainer.train_loop does every
        `check_every` steps, bench.py after its timed region."""
        if self._lsap_status is None:
            return
        bits = int(self._lsap_status.item())
        if bits:
            self._lsap_status.zero_()
        if bits & 1:
            raise ValueError("matrix contains invalid numeric entries (non-finite Hungarian matching cost: the step diverged)")
        if bits & 2:
            raise ValueError(f"a frame has more than {self.LSAP_DEVICE_MAX_G} targets: not solvable by the device LSAP")

    @staticmethod
    def solve(costs_host):
        out = []
        for C in costs_host:
            i, j = linear_sum_assignment(C)
            out.append((torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)))
        return out

    @torch.no_grad()
    def forward(self, outputs, targets, point_source=None):
        """Same contract as the reference: list of (index_i, index_j) per batch element."""
        costs = self.cost_matrices(outputs, targets, point_source)
        return self.solve([c.cpu().numpy() for c in costs])

    @torch.no_grad()
    def match_layers(self, layer_outputs, targets, point_source=None):
        """All decoder outputs at once: one D2H copy for len(layer_outputs) x frames cost matrices."""
        all_costs = [self.cost_matrices(o, targets, point_source) for o in layer_outputs]
        sizes = [[tuple(c.shape) for c in lc] for lc in all_costs]
        flat = torch.cat([c.reshape(-1) for lc in all_costs for c in lc]).cpu().numpy()  # the only sync
        res, off = [], 0
        for ls in sizes:
            host = []
            for (q, g) in ls:
                host.append(flat[off:off + q * g].reshape(q, g))
                off += q * g
            res.append(self.solve(host))
        return res

    def __repr__(self, _repr_indent=4):
        body = [f"cost_class: {self.cost_class}", f"cost_mask: {self.cost_mask}", f"cost_dice: {self.cost_dice}"]
        return "\n".join(["Matcher " + self.__class__.__name__] + [" " * _repr_indent + line for line in body])
