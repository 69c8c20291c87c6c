"""Independent PyTorch / numpy formulations used to cross-check the oracle's restatements of the
reference's CUDA-only operators (ROIAlign backward, ROIPool, deformable conv).  They are written
from the operator DEFINITIONS, vectorised and differentiable, i.e. not a
transcription of either the reference kernels or of oracle/detops_oracle.c.
"""
import math

import numpy as np
import torch


def _axis_table(start, bin_size, P, grid, size, dtype):
    """Per-axis sample table for ROIAlign: for p in [0,P), i in [0,grid):
    returns (low idx, high idx, w_low, w_high, valid) each of shape [P*grid]."""
    p = torch.arange(P, dtype=dtype).repeat_interleave(grid)
    i = torch.arange(grid, dtype=dtype).repeat(P)
    c = start + p * bin_size + (i + 0.5) * bin_size / grid
    valid = ~((c < -1.0) | (c > size))
    c = c.clamp(min=0)
    low = c.floor().long()
    edge = low >= size - 1
    low = torch.where(edge, torch.full_like(low, size - 1), low)
    high = torch.where(edge, low, low + 1)
    c = torch.where(edge, low.to(dtype), c)
    frac = c - low.to(dtype)
    return low, high, (1 - frac) * valid, frac * valid


def roi_align_torch(x, rois, scale, PH, PW, sampling_ratio):
    """x [N,C,H,W] (double), rois [K,5]; separable-matrix formulation:
    out[k,c] = Ay[k] @ x[b_k, c] @ Ax[k]^T / count."""
    N, C, H, W = x.shape
    outs = []
    for r in rois:
        b = int(r[0])
        sw, sh, ew, eh = [float(v) * scale for v in r[1:]]
        rw, rh = max(ew - sw, 1.0), max(eh - sh, 1.0)
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rh / PH))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(rw / PW))
        yl, yh, wyl, wyh = _axis_table(sh, rh / PH, PH, gh, H, x.dtype)
        xl, xh, wxl, wxh = _axis_table(sw, rw / PW, PW, gw, W, x.dtype)
        Ay = torch.zeros(PH * gh, H, dtype=x.dtype)
        Ay.scatter_add_(1, yl[:, None], wyl[:, None])
        Ay.scatter_add_(1, yh[:, None], wyh[:, None])
        Ax = torch.zeros(PW * gw, W, dtype=x.dtype)
        Ax.scatter_add_(1, xl[:, None], wxl[:, None])
        Ax.scatter_add_(1, xh[:, None], wxh[:, None])
        Ay = Ay.view(PH, gh, H).sum(1)
        Ax = Ax.view(PW, gw, W).sum(1)
        outs.append(torch.einsum("ph,chw,qw->cpq", Ay, x[b], Ax) / (gh * gw))
    if not outs:
        return x.new_zeros((0, C, PH, PW))
    return torch.stack(outs)


def _c_round(v):
    """C roundf: half away from zero."""
    return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)


def roi_pool_py(inp, rois, scale, PH, PW):
    N, C, H, W = inp.shape
    K = rois.shape[0]
    out = np.zeros((K, C, PH, PW), np.float32)
    amax = np.full((K, C, PH, PW), -1, np.int32)
    f32 = np.float32
    for k in range(K):
        b = int(rois[k, 0])
        x1, y1, x2, y2 = [_c_round(float(f32(v) * f32(scale))) for v in rois[k, 1:]]
        rw, rh = max(x2 - x1 + 1, 1), max(y2 - y1 + 1, 1)
        bh, bw = f32(rh) / f32(PH), f32(rw) / f32(PW)
        for ph in range(PH):
            hs = min(max(int(np.floor(f32(ph) * bh)) + y1, 0), H)
            he = min(max(int(np.ceil(f32(ph + 1) * bh)) + y1, 0), H)
            for pw in range(PW):
                ws = min(max(int(np.floor(f32(pw) * bw)) + x1, 0), W)
                we = min(max(int(np.ceil(f32(pw + 1) * bw)) + x1, 0), W)
                if he <= hs or we <= ws:
                    continue
                win = inp[b, :, hs:he, ws:we].reshape(C, -1)
                j = win.argmax(1)  # first maximum in row-major order
                out[k, :, ph, pw] = win[np.arange(C), j]
                amax[k, :, ph, pw] = (hs + j // (we - ws)) * W + ws + j % (we - ws)
    return out, amax


def deform_conv_torch(x, offset, mask, weight, bias, stride, pad, dil, group, dg):
    """Differentiable deformable conv (v1 when mask is None, v2 otherwise).
    x [B,C,H,W], offset [B,dg*2*k*k,Ho,Wo] (dh at 2*(i*kw+j), dw at +1), mask [B,dg*k*k,Ho,Wo]."""
    B, C, H, W = x.shape
    Cout, Cg, kh, kw = weight.shape
    Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    dt = x.dtype
    ho = torch.arange(Ho, dtype=dt).view(1, 1, Ho, 1)
    wo = torch.arange(Wo, dtype=dt).view(1, 1, 1, Wo)
    cols = []
    cpg = C // dg
    for i in range(kh):
        for j in range(kw):
            tap = i * kw + j
            per_g = []
            for g in range(dg):
                dh = offset[:, g * 2 * kh * kw + 2 * tap].unsqueeze(1)
                dw = offset[:, g * 2 * kh * kw + 2 * tap + 1].unsqueeze(1)
                hh = ho * stride - pad + i * dil + dh  # [B,1,Ho,Wo]
                ww = wo * stride - pad + j * dil + dw
                inside = (hh > -1) & (ww > -1) & (hh < H) & (ww < W)
                h0 = hh.detach().floor()
                w0 = ww.detach().floor()
                lh, lw = hh - h0, ww - w0
                xs = x[:, g * cpg:(g + 1) * cpg]
                val = 0
                for (hi, wi, wt) in ((h0, w0, (1 - lh) * (1 - lw)), (h0, w0 + 1, (1 - lh) * lw),
                                     (h0 + 1, w0, lh * (1 - lw)), (h0 + 1, w0 + 1, lh * lw)):
                    ok = (hi >= 0) & (hi <= H - 1) & (wi >= 0) & (wi <= W - 1) & inside
                    idx = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).long()  # [B,1,Ho,Wo]
                    v = xs.flatten(2).gather(2, idx.view(B, 1, -1).expand(B, cpg, -1))
                    val = val + v.view(B, cpg, Ho, Wo) * (wt * ok)
                if mask is not None:
                    val = val * mask[:, g * kh * kw + tap].unsqueeze(1)
                per_g.append(val)
            cols.append(torch.cat(per_g, 1))  # [B,C,Ho,Wo]
    col = torch.stack(cols, 2)  # [B,C,kh*kw,Ho,Wo]
    Mg = Cout // group
    outs = []
    for g in range(group):
        cg = col[:, g * Cg:(g + 1) * Cg].reshape(B, Cg * kh * kw, Ho * Wo)
        wg = weight[g * Mg:(g + 1) * Mg].reshape(Mg, Cg * kh * kw)
        outs.append(torch.einsum("mk,bkp->bmp", wg, cg))
    out = torch.cat(outs, 1).view(B, Cout, Ho, Wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out


def deform_psroi_pool_torch(data, rois, trans, no_trans, scale, output_dim, group_size, P, part_size,
                            S, trans_std):
    """Deformable PS-ROI pooling from its definition (Deformable ConvNets, eq. 5-6 as the reference
    discretises it): data [N,C,H,W] double, rois [K,5], trans [K,2*ncls,part,part] or None.
    Differentiable in `data` and `trans`; the border clamp is straight-through (the operator's
    offset gradient ignores the clamp).  Returns (out, count) [K,output_dim,P,P]."""
    N, C, H, W = data.shape
    dt = data.dtype
    ncls = 1 if no_trans else trans.shape[1] // 2
    cec = output_dim if no_trans else output_dim // ncls
    ar = torch.arange(P, dtype=dt)
    part = torch.floor(ar / P * part_size).long()
    grp = torch.floor(ar * group_size / P).long().clamp(0, group_size - 1)
    sub = torch.arange(S, dtype=dt)
    cls_of = torch.arange(output_dim) // cec
    # position-sensitive plane of (ctop, ph, pw)
    plane = (torch.arange(output_dim)[:, None, None] * group_size + grp[None, :, None]) * group_size + grp[None, None, :]
    outs, cnts = [], []
    for k in range(rois.shape[0]):
        b = int(rois[k, 0])
        x1, y1, x2, y2 = [float(_c_round(float(v))) for v in rois[k, 1:]]
        sw, sh = x1 * scale - 0.5, y1 * scale - 0.5
        ew, eh = (x2 + 1.0) * scale - 0.5, (y2 + 1.0) * scale - 0.5
        rw, rh = max(ew - sw, 0.1), max(eh - sh, 0.1)
        bw, bh = rw / P, rh / P
        if no_trans:
            tx = torch.zeros(output_dim, P, P, dtype=dt)
            ty = torch.zeros(output_dim, P, P, dtype=dt)
        else:
            t = trans[k].view(ncls, 2, part_size, part_size)[cls_of]          # [D,2,part,part]
            t = t[:, :, part][:, :, :, part]                                 # [D,2,P,P]
            tx, ty = t[:, 0] * trans_std, t[:, 1] * trans_std
        ws = ar[None, None, :] * bw + sw + tx * rw                            # [D,P,P]
        hs = ar[None, :, None] * bh + sh + ty * rh
        w = ws[..., None, None] + sub[None, None, None, None, :] * (bw / S)   # [D,P,P,S,S] (ih, iw)
        h = hs[..., None, None] + sub[None, None, None, :, None] * (bh / S)
        w, h = torch.broadcast_tensors(w, h)
        valid = ~((w < -0.5) | (w > W - 0.5) | (h < -0.5) | (h > H - 0.5))
        wc = w + (w.clamp(0, W - 1) - w).detach()
        hc = h + (h.clamp(0, H - 1) - h).detach()
        x0, x1i = wc.detach().floor().long(), wc.detach().ceil().long()
        y0, y1i = hc.detach().floor().long(), hc.detach().ceil().long()
        x0, x1i, y0, y1i = [v.clamp(0, m) for v, m in ((x0, W - 1), (x1i, W - 1), (y0, H - 1), (y1i, H - 1))]
        dx, dy = wc - x0.to(dt), hc - y0.to(dt)
        img = data[b].reshape(C, H * W)
        pl = plane[..., None, None].expand_as(x0)

        def tap(yy, xx):
            return img[pl, yy * W + xx]

        val = ((1 - dx) * (1 - dy) * tap(y0, x0) + (1 - dx) * dy * tap(y1i, x0)
               + dx * (1 - dy) * tap(y0, x1i) + dx * dy * tap(y1i, x1i))
        val = torch.where(valid, val, torch.zeros_like(val))
        cnt = valid.sum((-1, -2)).to(dt)
        outs.append(torch.where(cnt > 0, val.sum((-1, -2)) / cnt.clamp(min=1), torch.zeros_like(cnt)))
        cnts.append(cnt)
    if not outs:
        z = data.new_zeros((0, output_dim, P, P))
        return z, z
    return torch.stack(outs), torch.stack(cnts)
