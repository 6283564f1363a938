"""Loss helpers with the reference's names (loss.py) for code that assembles the loss
itself, e.g. the reference's main.py (main.py:439-562) running on top of
ta3n_amd.models.VideoModel.  They are a few elementwise ops on [B,C] / [B,2] logits.
The fused train step (ta3n_amd.engine.TrainEngine) does NOT use them: there the
whole loss assembly and its gradients are one HIP kernel (csrc/ta3n_pointwise.hip)."""
import torch
import torch.nn.functional as F


def cross_entropy_soft(pred):
    """loss.py:8-12."""
    return torch.mean(torch.sum(-F.softmax(pred, 1) * F.log_softmax(pred, 1), 1))


def attentive_entropy(pred, pred_domain):
    """loss.py:15-25: mean((1 + H(softmax(pred_domain))) * H(softmax(pred)))."""
    weights = 1 + torch.sum(-F.softmax(pred_domain, 1) * F.log_softmax(pred_domain, 1), 1)
    return torch.mean(weights * torch.sum(-F.softmax(pred, 1) * F.log_softmax(pred, 1), 1))


def dis_MCD(out1, out2):
    """loss.py:29-30."""
    return torch.mean(torch.abs(F.softmax(out1, dim=1) - F.softmax(out2, dim=1)))


class _GaussianKernelHip(torch.autograd.Function):
    """loss.py:46-59 on the GPU through the C ABI (ta3n_gaussian_kernel / ta3n_mmd_rowdiff, csrc/ta3n_mmd.hip): the [n, n] kernel
    matrix of the stacked rows in forward, the O(n^2 d) contraction back to the features in backward."""

    @staticmethod
    def forward(ctx, total, kernel_mul, kernel_num, fix_sigma):
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        total = total.contiguous()
        n, d = int(total.size(0)), int(total.size(1))
        k = torch.empty(n, n, dtype=torch.float32, device=total.device)
        kp = torch.empty_like(k)
        scratch = torch.empty(int(L.ta3n_gaussian_kernel_scratch_floats(n)), dtype=torch.float32, device=total.device)
        stream = C.c_void_p(torch.cuda.current_stream(total.device).cuda_stream)
        _lib.check(L.ta3n_gaussian_kernel(total.data_ptr(), n, d, float(kernel_mul), int(kernel_num), float(fix_sigma) if fix_sigma else 0.0,
                                          k.data_ptr(), kp.data_ptr(), scratch.data_ptr(), stream), "ta3n_gaussian_kernel")
        ctx.save_for_backward(total, kp)
        return k

    @staticmethod
    def backward(ctx, gk):
        import ctypes as C
        from . import _lib
        total, kp = ctx.saved_tensors
        n, d = int(total.size(0)), int(total.size(1))
        c = (2.0 * kp * (gk + gk.t())).contiguous()             # d ||t_p - t_q||^2 / d t_p = 2 (t_p - t_q), from both (p, q) and (q, p)
        out = torch.empty_like(total)
        stream = C.c_void_p(torch.cuda.current_stream(total.device).cuda_stream)
        _lib.check(_lib.lib().ta3n_mmd_rowdiff(c.data_ptr(), total.data_ptr(), n, d, 1.0, out.data_ptr(), stream), "ta3n_mmd_rowdiff")
        return out, None, None, None


def guassian_kernel(source, target, kernel_mul=2.0, kernel_num=5, fix_sigma=None):
    """loss.py:46-59 (name as in the reference): sum of kernel_num RBF kernels over the stacked [source; target] rows; the
    bandwidth is the mean pairwise squared distance (no gradient through it), scaled by kernel_mul^(i - kernel_num // 2)."""
    n = int(source.size(0)) + int(target.size(0))
    total = torch.cat([source, target], dim=0)
    if total.is_cuda and total.dtype == torch.float32 and n >= 2:      # the product path: HIP kernels (fp64 / CPU tensors: the torch form below)
        return _GaussianKernelHip.apply(total, kernel_mul, kernel_num, fix_sigma)
    # ||x_i - x_j||^2 in the reference's explicit difference form (loss.py:50-52), a block of rows at a time so the [rows, n, d]
    # intermediate stays small.  (Not |x|^2 + |y|^2 - 2 x.y: in fp32 that cancels catastrophically for near-duplicate rows -
    # zero-padded dummy rows, logits - and moves the data-dependent bandwidth)
    rows = max(1, min(n, (1 << 24) // max(n * int(total.size(1)), 1)))
    l2 = torch.cat([((total[r0:r0 + rows, None, :] - total[None, :, :]) ** 2).sum(2) for r0 in range(0, n, rows)], dim=0)
    bandwidth = fix_sigma if fix_sigma else torch.sum(l2.detach()) / (n * n - n)
    bandwidth = bandwidth / kernel_mul ** (kernel_num // 2)
    return sum(torch.exp(-l2 / (bandwidth * kernel_mul ** i)) for i in range(kernel_num))


def _quadrants(kernels, batch_size):
    XX, YY = kernels[:batch_size, :batch_size], kernels[batch_size:, batch_size:]
    XY, YX = kernels[:batch_size, batch_size:], kernels[batch_size:, :batch_size]
    return torch.mean(XX + YY - XY - YX)


def mmd_rbf(source, target, kernel_mul=2.0, kernel_num=5, fix_sigma=None, ver=2):
    """loss.py:61-85 (ver 2, the one main.py uses)."""
    if ver != 2:
        raise ValueError('ver == 2 (main.py:470, 497 never ask for the linear-time estimate)')
    return _quadrants(guassian_kernel(source, target, kernel_mul, kernel_num, fix_sigma), int(source.size(0)))


def JAN(source_list, target_list, kernel_muls=[2.0, 2.0], kernel_nums=[2, 5], fix_sigma_list=[None, None], ver=2):
    """loss.py:87-120 (ver 2): the layers' kernels multiplied, then the same four-quadrant mean."""
    if ver != 2:
        raise ValueError('ver == 2')
    joint = None
    for i in range(len(source_list)):
        k = guassian_kernel(source_list[i], target_list[i], kernel_muls[i], kernel_nums[i], fix_sigma_list[i])
        joint = k if joint is None else joint * k
    return _quadrants(joint, int(source_list[0].size(0)))

