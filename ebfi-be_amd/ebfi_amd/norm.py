"""GroupNorm on the gfx950 kernels of csrc/groupnorm.hip (drop-in for ``F.group_norm`` on contiguous
NCHW fp32 GPU tensors; the module keeps nn.GroupNorm as parameter container, so state_dict is unchanged).
Reference use: ExposureDecision applies one shared GroupNorm(4, 64) to two full-resolution maps
(models/Ours/model_singleframe.py:36,66-67)."""
import torch
import torch.nn.functional as F

from . import _native as N


class _GroupNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps):
        x = x.contiguous()
        B, C = x.shape[0], x.shape[1]
        HW = x.numel() // max(B * C, 1)
        lib = N.lib()
        y = torch.empty_like(x)
        mean = torch.empty(B * groups, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        need = int(lib.ebfi_groupnorm_workspace(B, C))
        ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        with torch.cuda.device_of(x):
            rc = lib.ebfi_groupnorm_forward(N.ptr(x), N.ptr(weight), N.ptr(bias), N.ptr(y), N.ptr(mean), N.ptr(rstd), B, C, HW,
                                            groups, float(eps), N.ptr(ws), need, N.stream_ptr(x.device))
        N.check(rc, "ebfi_groupnorm_forward")
        ctx.groups = groups
        ctx.save_for_backward(x, weight, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, mean, rstd = ctx.saved_tensors
        gy = gy.contiguous()
        B, C = x.shape[0], x.shape[1]
        HW = x.numel() // max(B * C, 1)
        lib = N.lib()
        gx = torch.empty_like(x)
        need_affine = weight is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        gw = torch.empty(C, dtype=torch.float32, device=x.device) if need_affine else None
        gb = torch.empty(C, dtype=torch.float32, device=x.device) if need_affine else None
        need = int(lib.ebfi_groupnorm_workspace(B, C)) + 8 * B * ctx.groups
        ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        with torch.cuda.device_of(x):
            rc = lib.ebfi_groupnorm_backward(N.ptr(gy), N.ptr(x), N.ptr(weight), N.ptr(mean), N.ptr(rstd), N.ptr(gx), N.ptr(gw),
                                             N.ptr(gb), B, C, HW, ctx.groups, N.ptr(ws), need, N.stream_ptr(x.device))
        N.check(rc, "ebfi_groupnorm_backward")
        return gx, gw, gb, None, None


def group_norm(x, module):
    """`module` is an nn.GroupNorm; uses the native kernels when the tensor qualifies."""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() >= 3 and (x.numel() // max(x.shape[0] * x.shape[1], 1)) % 4 == 0 \
            and not torch.is_autocast_enabled():
        return _GroupNormFn.apply(x, module.weight, module.bias, module.num_groups, module.eps)
    return F.group_norm(x, module.num_groups, module.weight, module.bias, module.eps)
