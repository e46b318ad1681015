"""Dense layers of the GAN feature-augmentation path (SURVEY 8 f-4 b) on this package's own kernels, with autograd.

The reference builds its generator and discriminators from nn.Linear / nn.Conv2d (augment/graphconv.py:157-176, augment/crn.py:64-142,
augment/gan.py:74-160) and lets cuBLAS / cuDNN run them.  Here every contraction of those networks is ONE kind of call: rows x weights^T
on the exact-fp32 MFMA GEMM of csrc/gemm.hip (`sgg_gemm`), for the convolutions after a patch-matrix pass (`sgg_im2col`); the backward
is the same GEMM on transposed operands (`sgg_transpose`), the input gradient of a convolution goes back through `sgg_col2im`, bias
gradients are fixed-order column sums (`sgg_colsum`).  Activations, normalisations and poolings around them are elementwise / small
reductions and stay torch expressions.

Layout: channels-last everywhere.  A feature map is a [B, H, W, C] tensor, i.e. the [B*H*W, C] row matrix the GEMM reads -- a 1x1
convolution needs no data movement at all, a k x k one a single patch-matrix pass.  The modules keep the parameter names and shapes of
the torch layers they stand for (weight [Cout, Cin, k, k] / [out, in], bias; weight_orig / weight_u / weight_v with spectral
normalisation), so the reference's checkpoints load by name.

The kernels run on the GPU only: a CPU tensor raises (there is no fallback path).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


# Arithmetic of every contraction in this file (process-wide; set_compute):
#   'f32' (default) exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): any magnitude, the golden vectors' tolerances;
#   'x3'  f16 split operands x = hi + lo, products hi.hi + hi.lo + lo.hi accumulated in fp32 on the 16-bit matrix cores (ops.gemm's
#         split-operand lowering, csrc/util.hip sgg_split3): ~1e-6 relative of exact fp32 -- for operands INSIDE f16's range only (an
#         unnormalised discriminator whose activations pass 65504 overflows the hi half: measured, tools/dbg_gan.py), and at the GAN's
#         shapes the split passes cost what the faster products gain (bench.py --mode gqa_gan: 72 ms either way) -- opt-in;
#   'f16' operands rounded to f16 once, fp32 accumulate (the SGG head's default mode; for the GAN an accuracy trade the caller opts into).
_COMPUTE = ['f32']


def set_compute(mode):
    assert mode in ('x3', 'f32', 'f16'), mode
    prev = _COMPUTE[0]
    _COMPUTE[0] = mode
    return prev


def compute_mode():
    return _COMPUTE[0]


def product(x, w, bias=None):
    """x [M,Kp] f32 . w [N,Kp]^T (+ bias) -> [M,N] f32 in the file's arithmetic (Kp a multiple of 32, zero-padded)"""
    mode = _COMPUTE[0]
    if mode == 'f16':
        x16, w16 = ops.cast(x, torch.float16), ops.cast(w, torch.float16)
        if x16.shape[1] % 64:
            x16, w16 = F.pad(x16, (0, 64 - x16.shape[1] % 64)), F.pad(w16, (0, 64 - w16.shape[1] % 64))
        return ops.gemm(x16, w16, bias, out_dtype=torch.float32)
    if mode == 'x3':
        # the split operands are 3 K f16 columns per row: a long reduction (a weight gradient: K = the rows of the layer's input) is cut into
        # K ranges that stay under the kernels' 4 GiB operand span, partial products summed in fp32 (in range order: deterministic)
        rows = max(x.shape[0], w.shape[0])
        kmax = max(64, (ops.SPAN_LIMIT // (rows * 6)) // 64 * 64)
        if x.shape[1] > kmax:
            out = None
            for k0 in range(0, x.shape[1], kmax):
                part = ops.gemm(x[:, k0:k0 + kmax], w[:, k0:k0 + kmax], bias if k0 == 0 else None, out_dtype=torch.float32, x3=True)
                out = part if out is None else out.add_(part)
            return out
    return ops.gemm(x, w, bias, out_dtype=torch.float32, x3=(mode == 'x3'))


def _need_gpu(x):
    if not x.is_cuda:
        raise RuntimeError('sgg_amd.dense: the dense layers run on the HIP kernels only (got a %s tensor)' % x.device.type)


def _cols32(x):
    """[M,K] f32 -> contiguous [M, K rounded up to 32] with zero columns (what the fp32 MFMA kernel's K tile takes)."""
    M, K = x.shape
    Kp = (K + 31) // 32 * 32
    if Kp == K and x.is_contiguous() and x.dtype == torch.float32:
        return x
    buf = x.new_zeros((M, Kp), dtype=torch.float32)
    buf[:, :K].copy_(x)
    return buf


class _Affine(torch.autograd.Function):
    """y[M,N] = x[M,Kx] . w[N,K]^T (+ b);  Kx >= K, columns of x past K must be zero (a patch matrix arrives padded)."""

    @staticmethod
    def forward(ctx, x, w, b):
        _need_gpu(x)
        xp, wp = _cols32(x.detach()), _cols32(w.detach())
        if xp.shape[1] != wp.shape[1]:
            raise ValueError('dense: %d input columns against %d weight columns' % (x.shape[1], w.shape[1]))
        ctx.save_for_backward(xp, wp)
        ctx.dims = (x.shape[1], w.shape[0], w.shape[1], b is not None)
        if xp.shape[0] == 0:
            return xp.new_zeros((0, w.shape[0]))
        return product(xp, wp, b.detach().float().contiguous() if b is not None else None)

    @staticmethod
    def backward(ctx, dy):
        xp, wp = ctx.saved_tensors
        Kx, N, K, has_b = ctx.dims
        M = xp.shape[0]
        dx = dw = db = None
        if M == 0:
            return (xp.new_zeros((0, Kx)) if ctx.needs_input_grad[0] else None, wp.new_zeros((N, K)) if ctx.needs_input_grad[1] else None,
                    wp.new_zeros(N) if has_b and ctx.needs_input_grad[2] else None)
        dy = dy.contiguous().float()
        if ctx.needs_input_grad[0]:
            wt = ops.transpose(wp, pad_to=32)                           # [Kp, N32]
            dx = product(_cols32(dy), wt)[:, :Kx]
        if ctx.needs_input_grad[1]:
            # dW = dy^T x: both operands transposed once (zero-padded reduction rows), then the same GEMM
            dw = product(ops.transpose(dy), ops.transpose(xp))[:, :K]
        if has_b and ctx.needs_input_grad[2]:
            db = ops.colsum(dy)
        return dx, dw, db


def affine(x, weight, bias=None):
    """x [..., K] -> [..., N] = x . weight^T + bias (weight [N,K]) on sgg_gemm; differentiable in all three."""
    lead = x.shape[:-1]
    y = _Affine.apply(x.reshape(-1, x.shape[-1]), weight, bias)
    return y.view(*lead, weight.shape[0])


class _Patches(torch.autograd.Function):
    """x [B,H,W,C] -> patch matrix [B*Ho*Wo, (k*k*C) rounded up to 32], columns (ky, kx, c): sgg_im2col / sgg_col2im."""

    @staticmethod
    def forward(ctx, x, k, stride, pad):
        _need_gpu(x)
        x = x.detach().contiguous().float()
        ctx.geom = (tuple(x.shape), k, stride, pad)
        cols, _, _ = ops.im2col(x, k, stride, pad, Kp=(k * k * x.shape[3] + 31) // 32 * 32)
        return cols

    @staticmethod
    def backward(ctx, d_cols):
        shape, k, stride, pad = ctx.geom
        return ops.col2im(d_cols.contiguous().float(), shape, k, stride, pad), None, None, None


def conv2d(x, weight, bias=None, padding=0, stride=1):
    """Channels-last convolution: x [B,H,W,Cin], weight [Cout,Cin,k,k] (torch's layout: checkpoints load) -> [B,Ho,Wo,Cout]."""
    B, H, W, C = x.shape
    Cout, Cin, k, k2 = weight.shape
    assert Cin == C and k == k2, (tuple(x.shape), tuple(weight.shape))
    Ho, Wo = (H + 2 * padding - k) // stride + 1, (W + 2 * padding - k) // stride + 1
    if k == 1 and padding == 0 and stride == 1:
        rows, wmat = x.reshape(-1, C), weight.reshape(Cout, C)
    else:
        rows = _Patches.apply(x, k, stride, padding)
        wmat = weight.permute(0, 2, 3, 1).reshape(Cout, k * k * C)       # columns in the patch matrix's (ky, kx, c) order
        if rows.shape[1] != wmat.shape[1]:
            wmat = F.pad(wmat, (0, rows.shape[1] - wmat.shape[1]))
    return _Affine.apply(rows, wmat, bias).view(B, Ho, Wo, Cout)


# ------------------------------------------------------------------------------------------------ modules
class Linear(nn.Linear):
    """nn.Linear's parameters and initialisation; the product runs on sgg_gemm."""

    def forward(self, x):
        return affine(x, self.weight, self.bias)


def l2_normalize(v, eps=1e-12):
    return v / v.norm().clamp_min(eps)


class Conv2d(nn.Module):
    """A channels-last convolution layer with nn.Conv2d's parameter names, shapes and default initialisation.  spectral=True: the
    weight is divided by its largest singular value, estimated by one power-iteration step per training forward on the persistent
    vectors weight_u / weight_v -- the scheme and the state names of torch.nn.utils.spectral_norm (augment/gan.py:77-79), with the two
    matrix-vector products written as broadcast-multiply + sum (a weight-preparation step, no library call)."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0, spectral=False, eps=1e-12):
        super(Conv2d, self).__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.padding = in_channels, out_channels, kernel_size, padding
        self.spectral, self.eps = spectral, eps
        ref = nn.Conv2d(in_channels, out_channels, kernel_size, padding=padding)     # torch's default initialisation, then dropped
        self.bias = nn.Parameter(ref.bias.detach().clone())
        if spectral:
            self.weight_orig = nn.Parameter(ref.weight.detach().clone())
            mat = self.weight_orig.detach().reshape(out_channels, -1)
            self.register_buffer('weight_u', l2_normalize(torch.randn(mat.shape[0]), eps))
            self.register_buffer('weight_v', l2_normalize(torch.randn(mat.shape[1]), eps))
        else:
            self.weight = nn.Parameter(ref.weight.detach().clone())

    def effective_weight(self):
        if not self.spectral:
            return self.weight
        w = self.weight_orig
        mat = w.reshape(w.shape[0], -1)
        if self.training:
            with torch.no_grad():
                m = mat.detach()
                self.weight_v.copy_(l2_normalize((m * self.weight_u[:, None]).sum(0), self.eps))      # v <- W^T u / |.|
                self.weight_u.copy_(l2_normalize((m * self.weight_v[None, :]).sum(1), self.eps))      # u <- W v / |.|
        u, v = self.weight_u.clone(), self.weight_v.clone()
        sigma = (u * (mat * v[None, :]).sum(1)).sum()
        return w / sigma

    def forward(self, x):
        return conv2d(x, self.effective_weight(), self.bias, padding=self.padding)

    def extra_repr(self):
        return '%d, %d, kernel_size=%d, padding=%d%s' % (self.in_channels, self.out_channels, self.kernel_size, self.padding,
                                                       ', spectral' if self.spectral else '')


class BatchNormRows(nn.modules.batchnorm._BatchNorm):
    """Batch normalisation over every axis but the LAST one (channels-last feature maps [B,H,W,C] and row matrices [M,C] alike) with
    BatchNorm1d / BatchNorm2d's parameters, buffers and update rule (biased variance to normalise, unbiased into running_var).
    Being a _BatchNorm it is what torch.nn.SyncBatchNorm.convert_sync_batchnorm looks for; see RowsSyncBatchNorm below."""

    def _check_input_dim(self, x):
        if x.dim() < 2:
            raise ValueError('expected at least 2D input (got %dD)' % x.dim())

    def forward(self, x):
        C = x.shape[-1]
        rows = x.reshape(-1, C)
        if self.training or not self.track_running_stats:
            n = rows.shape[0]
            mean = rows.mean(0)
            var = (rows - mean).square().mean(0)
            if self.training and self.track_running_stats:
                with torch.no_grad():
                    self.num_batches_tracked += 1
                    mom = self.momentum if self.momentum is not None else 1.0 / float(self.num_batches_tracked)
                    self.running_mean.mul_(1 - mom).add_(mean.detach(), alpha=mom)
                    self.running_var.mul_(1 - mom).add_(var.detach() * (n / max(n - 1, 1)), alpha=mom)
        else:
            mean, var = self.running_mean, self.running_var
        scale = torch.rsqrt(var + self.eps)
        if self.affine:
            scale = scale * self.weight
        y = (rows - mean) * scale
        if self.affine:
            y = y + self.bias
        return y.view(x.shape)


class RowsSyncBatchNorm(nn.SyncBatchNorm):
    """SyncBatchNorm for channels-last inputs: torch's module wants the channel axis second, so the rows are handed over as [M,C]
    (which it accepts) and the result is viewed back."""

    def forward(self, x):
        return super(RowsSyncBatchNorm, self).forward(x.reshape(-1, x.shape[-1])).view(x.shape)


def sync_batchnorm_(module):
    """torch.nn.SyncBatchNorm.convert_sync_batchnorm for models built from this file's layers: BatchNormRows -> RowsSyncBatchNorm
    (same parameters and buffers), in place; returns the module."""
    for name, child in list(module.named_children()):
        if isinstance(child, BatchNormRows):
            new = RowsSyncBatchNorm(child.num_features, child.eps, child.momentum, child.affine, child.track_running_stats)
            if child.affine:
                new.weight, new.bias = child.weight, child.bias
            new.running_mean, new.running_var, new.num_batches_tracked = child.running_mean, child.running_var, child.num_batches_tracked
            new.training = child.training
            setattr(module, name, new)
        else:
            sync_batchnorm_(child)
    return module
