"""Thin Python wrappers over the C ABI: one function per entry point, device tensors in/out.

Activations are "padded NHWC" tensors [N, H+2, W+2, C] (zero halo) — see include/vnqa_hip.h.
"""
import ctypes
import os

import torch

from . import _lib as L


def pack_conv_weight(w_oihw, dtype, out_scale=None, transpose_flip=False, c_out_pad=None, c_in_pad=None):
    """OIHW fp32 -> K-major [rows][taps][ch] in `dtype` (rows=c_out_pad, or c_in_pad if flipped)."""
    w = w_oihw.detach().float().contiguous()
    c_out, c_in = w.shape[0], w.shape[1]
    taps = w[0, 0].numel()                      # 1, 9 (OIHW) or 27 (OIDHW)
    c_out_pad = c_out_pad or L.round_up(c_out, 64)
    c_in_pad = c_in_pad or L.round_up(c_in, 64)
    rows, kch = (c_in_pad, c_out_pad) if transpose_flip else (c_out_pad, c_in_pad)
    wt = torch.empty((rows, taps, kch), dtype=dtype, device=w.device)
    sc = out_scale.detach().float().contiguous() if out_scale is not None else None
    L.check(L.lib().vnqa_pack_conv_weight(L.ptr(w), c_out, c_in, taps, c_out_pad, c_in_pad, L.ptr(sc),
                                          1 if transpose_flip else 0, L.dtype_id(dtype), L.ptr(wt),
                                          L.stream()), "vnqa_pack_conv_weight")
    return wt


def pad_vec(v, n):
    """fp32 per-channel vector zero-padded to n entries (the vector itself when nothing has to change)."""
    if v.numel() == n and v.dtype == torch.float32 and v.is_contiguous():
        return v.detach()
    out = torch.zeros(n, dtype=torch.float32, device=v.device)
    out[: v.numel()] = v.detach().float()
    return out


class TiledWeight(object):
    """Pre-tiled conv weights (vnqa_pack_conv_weight_tiled): opaque LDS-image buffer + its geometry."""

    def __init__(self, data, c_out, taps, c_in_pad, tile):
        self.data, self.c_out, self.taps, self.c_in_pad, self.tile = data, c_out, taps, c_in_pad, tile


def pack_conv_weight_tiled(w_oihw, dtype, tile, out_scale=None, c_out_pad=None, c_in_pad=None):
    w = w_oihw.detach().float().contiguous()
    c_out, c_in, kh, kw = w.shape
    taps = kh * kw
    c_out_pad = c_out_pad or L.round_up(c_out, 64)
    c_in_pad = c_in_pad or L.round_up(c_in, 64)
    did = L.dtype_id(dtype)
    nbytes = L.lib().vnqa_conv_weight_tiled_bytes(c_out_pad, c_in_pad, taps, tile, did)
    assert nbytes > 0, "tile %d has no tiled weight layout" % tile
    buf = torch.empty(nbytes // (2 if L.is_half(dtype) else 4), dtype=dtype, device=w.device)
    sc = out_scale.detach().float().contiguous() if out_scale is not None else None
    L.check(L.lib().vnqa_pack_conv_weight_tiled(L.ptr(w), c_out, c_in, taps, c_in_pad, L.ptr(sc), tile, did,
                                                L.ptr(buf), L.stream()), "vnqa_pack_conv_weight_tiled")
    return TiledWeight(buf, c_out_pad, taps, c_in_pad, tile)


def empty_padded(shape, dtype, device):
    """Fresh padded-NHWC buffer [N,Hp,Wp,C] with a ZERO 1-pixel halo and an uninitialised interior (for kernels that
    write every interior pixel): 4x less fill traffic than torch.zeros on the 16x16 trunk maps."""
    out = torch.empty(shape, dtype=dtype, device=device)
    N, Hp, Wp, C = shape
    L.check(L.lib().vnqa_zero_halo(L.ptr(out), N, Hp, Wp, C, L.dtype_id(dtype), L.stream()), "vnqa_zero_halo")
    return out


# ---- split operands (precision 'fp16h', csrc/split3.hip) ---------------------------------------------------------------------------
# A 16-bit MFMA operand carries 11 significand bits.  Where the logits-error budget says a rounding matters (profiles/
# r05_precision_budget*.txt) the operand is SPLIT instead: v = hi + lo with hi = h16(v), lo = h16(v - hi) (22 bits), and the
# contraction runs as two or three products of 16-bit halves with fp32 accumulation —
#   weights only   x . w_hi + x . w_lo              the activation is read twice along K by the igemm's wrap variant
#                                                   (VNQA_CONV_X_WRAP2 / VNQA_GEMM_X_WRAP2: no copy), `split_weights=True`
#   both operands  x_hi w_hi + x_lo w_hi + x_hi w_lo   the producer lays the activation out as [hi | lo | hi] (VNQA_CONV_DUAL_OUT |
#                                                   VNQA_CONV_DUAL_HI2), the consumer is a PLAIN conv over 3 C channels against
#                                                   [w_hi | w_hi | w_lo], `split_in=True`
# (x_lo . w_lo, 2^-22 relative, is dropped.)  The split weights are made from the fp32 K-major pack in one pass and cached on it.
def _split_weight(wt, parts):
    """fp32 K-major weights [..][k] -> 16-bit [..][len(parts) * k], the halves named in `parts` ('h' / 'l') along the innermost axis,
    in ONE pass (vnqa_split3_f32)."""
    k = wt.shape[-1]
    assert wt.dtype == torch.float32 and wt.is_contiguous() and k % 8 == 0 and parts in ("hhl", "hl")
    rows = wt.numel() // k
    out = torch.empty(wt.shape[:-1] + (len(parts) * k,), dtype=L.half_dtype(), device=wt.device)
    base, es = out.data_ptr(), out.element_size()
    hi, hi2, lo = (base, base + k * es, base + 2 * k * es) if parts == "hhl" else (base, None, base + k * es)
    L.check(L.lib().vnqa_split3_f32(L.ptr(wt), ctypes.c_void_p(hi), ctypes.c_void_p(lo), ctypes.c_void_p(hi2) if hi2 else None,
                                    rows, k, k, len(parts) * k, None, L.stream()), "vnqa_split3_f32(weights)")
    return out


def _cached_split(wt, parts, slot):
    cached = getattr(wt, slot, None)
    if cached is not None and cached[0] == wt._version:
        return cached[1]
    w = _split_weight(wt, parts)
    try:
        setattr(wt, slot, (wt._version, w))
    except (AttributeError, RuntimeError):
        pass
    return w


def split_weight3(wt):
    """[w_hi | w_hi | w_lo] along the contraction axis — the weight operand of a three-product conv over a [hi | lo | hi] activation;
    cached on the tensor object for packs that persist (re-made when the tensor was modified in place)."""
    return _cached_split(wt, "hhl", "_vnqa_w3")


def split_weight2(wt):
    """[w_hi | w_lo] along the contraction axis — the weight operand of a two-product conv / GEMM over a plain 16-bit activation."""
    return _cached_split(wt, "hl", "_vnqa_w2")


def conv2d_igemm(x, wt, bias=None, relu=False, pool2=False, post_scale=None, post_shift=None,
                 x_halo=1, y_halo=1, out=None, tile=L.TILE_AUTO, border_sub=None, desc_flags=0,
                 split_in=False, split_weights=False, dual_out=False, f32_epilogue=False, relu_floor=None):
    """x: padded NHWC [N,H+2h,W+2h,Cin]; wt: [Cout][taps][Cin] or a TiledWeight; returns padded NHWC output.
    Precision 'fp16h' (see the block comment above):
      split_weights — x a plain 16-bit tensor, wt the fp32 K-major pack: x . w_hi + x . w_lo on the igemm's wrap variant;
      split_in      — x a [hi | lo | hi] tensor (3 Cin channels), wt the fp32 pack: the plain conv over 3 Cin channels against
                      [w_hi | w_hi | w_lo], any tile;
      dual_out      — 2 / 3: the output as [hi | lo] / [hi | lo | hi] (patch-stationary tiles; the 256x256 implicit-GEMM tiles for the
                      geometries those do not serve and for the composed 5x5 conv: the same bits)."""
    if split_in:
        assert L.is_half(x.dtype) and wt.dtype == torch.float32 and not isinstance(wt, TiledWeight) and x.shape[-1] == 3 * wt.shape[2] \
            and not split_weights
        wt = split_weight3(wt)
    N, Hp, Wp, Cin = x.shape
    H, W = Hp - 2 * x_halo, Wp - 2 * x_halo
    tiled = isinstance(wt, TiledWeight)
    if split_weights:          # two products: x read twice along K against [w_hi | w_lo] (VNQA_CONV_X_WRAP2)
        assert L.is_half(x.dtype) and not tiled and wt.dtype == torch.float32 and wt.shape[2] == Cin and Cin % 64 == 0
        wt = split_weight2(wt)
        desc_flags = int(desc_flags) | L.CONV_X_WRAP2
        if border_sub is not None and border_sub.dtype != x.dtype:       # (the kernel subtracts it in the output's element type)
            border_sub = border_sub.to(x.dtype).contiguous()
        if tile not in (L.TILE_AUTO, L.TILE_256x256, L.TILE_256x128, L.TILE_256x64, L.TILE_STEM_256x256, 15, 17):
            tile = L.TILE_AUTO
    if tiled:
        c_out, taps, cin_w, tile = wt.c_out, wt.taps, wt.c_in_pad, wt.tile
        wt = wt.data
    else:
        c_out, taps, cin_w = wt.shape
    assert cin_w == (2 * Cin if split_weights else Cin) and wt.dtype == x.dtype, (cin_w, x.shape, wt.dtype, x.dtype)
    Ho, Wo = (H // 2, W // 2) if pool2 else (H, W)
    flags = 0
    if dual_out:        # 2 (or True): [hi | lo]; 3: [hi | lo | hi]
        segs = 3 if int(dual_out) == 3 else 2
        # patch-stationary tiles (3x3, no border correction) or — geometries they do not serve, the composed 5x5 — the 256x256 igemm tiles
        assert y_halo == 1 and c_out % 8 == 0 and (
            (tile in (L.TILE_PS_224x256, L.TILE_STEM_PS_224x256) and border_sub is None) or tile in (L.TILE_256x256, L.TILE_STEM_256x256)), tile
        if out is None:
            out = empty_padded((N, Ho + 2, Wo + 2, segs * c_out), x.dtype, x.device)
        assert out.shape[-1] == segs * c_out
        flags = L.CONV_DUAL_OUT | (L.CONV_DUAL_HI2 if segs == 3 else 0)
    elif f32_epilogue:      # one value, pool / affine in fp32, ONE rounding (VNQA_CONV_F32_EPILOGUE; the dual epilogue's hi half);
        # f32_epilogue == 2: written twice, [v | v] — the operand of a two-product consumer with split WEIGHTS [w_hi | w_lo]
        assert y_halo == 1 and c_out % 8 == 0 and L.is_half(x.dtype) and (
            (tile in (L.TILE_PS_224x256, L.TILE_STEM_PS_224x256) and border_sub is None) or tile in (L.TILE_256x256, L.TILE_STEM_256x256)), tile
        twin = int(f32_epilogue) == 2
        if out is None:
            out = empty_padded((N, Ho + 2, Wo + 2, (2 if twin else 1) * c_out), x.dtype, x.device)
        assert out.shape[-1] >= (2 if twin else 1) * c_out
        flags = L.CONV_F32_EPILOGUE | (L.CONV_DUAL_HI2 if twin else 0)
    if out is None:
        if y_halo == 1 and c_out % 8 == 0 and tile not in (11, 12, 20, 21):
            # fresh output: the kernel zeroes the halo ring itself (VNQA_CONV_ZERO_HALO), no fill / halo launch
            out = torch.empty((N, Ho + 2, Wo + 2, c_out), dtype=x.dtype, device=x.device)
            flags = L.CONV_ZERO_HALO
        elif y_halo == 1 and c_out % 8 == 0:
            out = empty_padded((N, Ho + 2, Wo + 2, c_out), x.dtype, x.device)
        else:
            out = torch.zeros((N, Ho + 2 * y_halo, Wo + 2 * y_halo, c_out), dtype=x.dtype, device=x.device)
    if relu_floor is not None:      # VNQA_CONV_RELU_FLOOR: the per-channel floor travels in the post_shift slot (the stem's tiles)
        assert post_scale is None and post_shift is None and relu and not dual_out and not f32_epilogue and tile in (L.TILE_STEM_256x256, L.TILE_STEM_PS_224x256)
        post_shift = relu_floor
        flags |= L.CONV_RELU_FLOOR
    d = L.ConvDesc(L.dtype_id(x.dtype), N, H, W, cin_w, c_out, out.shape[-1], taps, x_halo, y_halo,
                   int(relu), 1 if pool2 else 0, tile, 1 if tiled else 0, 0, flags | int(desc_flags))
    L.check(L.lib().vnqa_conv2d_igemm_fwd_ex(ctypes.byref(d), L.ptr(x), L.ptr(wt), L.ptr(bias), L.ptr(post_scale),
                                             L.ptr(post_shift), L.ptr(border_sub), L.ptr(out), L.stream()),
            "vnqa_conv2d_igemm_fwd")
    return out


def _conv_desc(x, c_out, c_y, taps, relu, tile=L.TILE_AUTO):
    N, Hp, Wp, Cin = x.shape
    # (fused trunk convs write fresh outputs: the kernel zeroes their halo ring, VNQA_CONV_ZERO_HALO)
    return L.ConvDesc(L.dtype_id(x.dtype), N, Hp - 2, Wp - 2, Cin, c_out, c_y, taps, 1, 1, int(relu), 0, tile, 0, 0,
                      L.CONV_ZERO_HALO)


def conv2d_igemm_bnstats(x, wt, bias, relu, frame_of_i32, frame_off_i32, n_frames, min_frame_images, split_in=False):
    """conv (+bias, +ReLU) with the per-frame BatchNorm statistics of its output taken in the epilogue
    (vnqa_conv2d_igemm_fused_fwd, VNQA_EPI_BNSTATS).  Returns (y, mean [F,Cout], var [F,Cout]) or None when the frames are
    too small for the tile (the caller then runs conv2d_igemm + frame_bn_stats).
    split_in (precision 'fp16h'): x is a [hi | lo | hi] tensor, wt the fp32 pack — see conv2d_igemm."""
    N, Hp, Wp, _ = x.shape
    if split_in:
        assert wt.dtype == torch.float32 and x.shape[-1] == 3 * wt.shape[2]
        wt = split_weight3(wt)
    c_out, taps, _ = wt.shape
    d = _conv_desc(x, c_out, c_out, taps, relu)
    ws_bytes = L.lib().vnqa_conv2d_bnstats_workspace(ctypes.byref(d), int(min_frame_images))
    if ws_bytes < 0:
        return None
    y = torch.empty((N, Hp, Wp, c_out), dtype=x.dtype, device=x.device)
    ws = workspace(ws_bytes, x.device)
    mean = torch.empty((n_frames, c_out), dtype=torch.float32, device=x.device)
    var = torch.empty((n_frames, c_out), dtype=torch.float32, device=x.device)
    e = L.ConvEpilogue(kind=L.EPI_BNSTATS, n_frames=n_frames, min_frame_images=int(min_frame_images),
                       frame_of=frame_of_i32.data_ptr(), frame_off=frame_off_i32.data_ptr(), partial=ws.data_ptr(),
                       mean=mean.data_ptr(), var=var.data_ptr())
    L.check(L.lib().vnqa_conv2d_igemm_fused_fwd(ctypes.byref(d), L.ptr(x), L.ptr(wt), L.ptr(bias), ctypes.byref(e),
                                                L.ptr(y), L.stream()), "vnqa_conv2d_igemm_fused_fwd(BNSTATS)")
    return y, mean, var


def conv2d_igemm_split_out(x, wt, bias, relu, split_in=False, tile=L.TILE_PS_224x256):
    """conv (+bias, +ReLU) on the patch-stationary tile (or, where it does not serve the geometry, tile = TILE_256x256) with its fp32
    result kept as TWO plain 16-bit tensors (VNQA_EPI_SPLIT_OUT):
    returns (hi, lo), hi = h16(v), lo = h16(v - hi), both padded NHWC with a zero halo.  split_in: see conv2d_igemm."""
    N, Hp, Wp, _ = x.shape
    if split_in:
        assert wt.dtype == torch.float32 and x.shape[-1] == 3 * wt.shape[2]
        wt = split_weight3(wt)
    c_out, taps, _ = wt.shape
    assert L.is_half(x.dtype) and wt.dtype == x.dtype and taps == 9
    d = L.ConvDesc(L.dtype_id(x.dtype), N, Hp - 2, Wp - 2, x.shape[-1], c_out, c_out, taps, 1, 1, int(relu), 0, tile, 0, 0, 0)
    both = empty_padded((2 * N, Hp, Wp, c_out), x.dtype, x.device)
    hi, lo = both[:N], both[N:]
    e = L.ConvEpilogue(kind=L.EPI_SPLIT_OUT, y2=lo.data_ptr())
    L.check(L.lib().vnqa_conv2d_igemm_fused_fwd(ctypes.byref(d), L.ptr(x), L.ptr(wt), L.ptr(bias), ctypes.byref(e),
                                                L.ptr(hi), L.stream()), "vnqa_conv2d_igemm_fused_fwd(SPLIT_OUT)")
    return hi, lo


def conv_ps_supported(n, h, w, c_in, c_out, taps=9, pool2=False):
    """The library's own answer (vnqa_conv_ps_supported) to: do the patch-stationary tiles serve this 16-bit conv geometry?"""
    d = L.ConvDesc(L.BF16, n, h, w, c_in, c_out, c_out, taps, 2 if taps == 25 else 1, 1, 0, 1 if pool2 else 0, L.TILE_PS_224x256, 0, 0, 0)
    return bool(L.lib().vnqa_conv_ps_supported(ctypes.byref(d)))


def ps_fused_tile(x):
    """TILE_PS_224x256 when the patch-stationary kernel can run a 3x3 conv with a fused FILM_RES / ADD_MASK epilogue on this
    padded-NHWC input (asked of the library: vnqa_conv_ps_supported), else TILE_AUTO."""
    if not L.is_half(x.dtype):
        return L.TILE_AUTO
    N, Hp, Wp, C = x.shape
    return L.TILE_PS_224x256 if (C % 64 == 0 and conv_ps_supported(N, Hp - 2, Wp - 2, C, C)) else L.TILE_AUTO


def conv2d_igemm_film_res(x, wt, bias, gamma, beta, film_c, res, tile=L.TILE_AUTO, keep_z=True):
    """z = conv(x) + bias and out = relu(gamma[n] * z + beta[n]) + res in ONE launch (VNQA_EPI_FILM_RES).  gamma / beta:
    fp32 2-D views [n_img, >= film_c] with unit column stride (column slices of the FiLM generator's output).
    keep_z=False (inference): z, which only the backward reads, is not stored; returns (None, out)."""
    N, Hp, Wp, _ = x.shape
    c_out, taps, _ = wt.shape
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32 and gamma.stride(1) == 1 and beta.stride(1) == 1
    assert gamma.stride(0) == beta.stride(0) and res.shape == (N, Hp, Wp, c_out) and res.dtype == x.dtype
    d = _conv_desc(x, c_out, c_out, taps, False, tile if taps == 9 else L.TILE_AUTO)
    z = torch.empty((N, Hp, Wp, c_out), dtype=x.dtype, device=x.device) if keep_z else None
    out = torch.empty((N, Hp, Wp, c_out), dtype=x.dtype, device=x.device)
    e = L.ConvEpilogue(kind=L.EPI_FILM_RES, film_ld=gamma.stride(0), film_c=int(film_c), gamma=gamma.data_ptr(),
                       beta=beta.data_ptr(), res=res.data_ptr(), y2=out.data_ptr())
    L.check(L.lib().vnqa_conv2d_igemm_fused_fwd(ctypes.byref(d), L.ptr(x), L.ptr(wt), L.ptr(bias), ctypes.byref(e),
                                                L.ptr(z), L.stream()), "vnqa_conv2d_igemm_fused_fwd(FILM_RES)")
    return z, out


def conv2d_igemm_add_mask(x, wt, add, mask_src, tile=L.TILE_AUTO):
    """(conv(x) + add) * [mask_src > 0] in ONE launch (VNQA_EPI_ADD_MASK): the dgrad of the FiLM block's 3x3 conv joined with the
    residual branch's gradient and masked by the 1x1 conv's ReLU."""
    N, Hp, Wp, _ = x.shape
    c_out, taps, _ = wt.shape
    assert add.shape == (N, Hp, Wp, c_out) and mask_src.shape == add.shape and add.dtype == x.dtype == mask_src.dtype
    d = _conv_desc(x, c_out, c_out, taps, False, tile if taps == 9 else L.TILE_AUTO)
    y = torch.empty((N, Hp, Wp, c_out), dtype=x.dtype, device=x.device)
    e = L.ConvEpilogue(kind=L.EPI_ADD_MASK, res=add.data_ptr(), y2=mask_src.data_ptr())
    L.check(L.lib().vnqa_conv2d_igemm_fused_fwd(ctypes.byref(d), L.ptr(x), L.ptr(wt), None, ctypes.byref(e), L.ptr(y),
                                                L.stream()), "vnqa_conv2d_igemm_fused_fwd(ADD_MASK)")
    return y


def film_relu_res_bwd_ld(dout, z, gamma, beta, film_c, dgamma, dbeta):
    """FiLM backward on column-slice views: gamma/beta [n_img, >= film_c] (row stride = view stride), gradients written into
    the views dgamma/dbeta (same column range of the gradient matrix)."""
    N, hp, wp, c = z.shape
    assert gamma.stride(1) == 1 and dgamma.stride(1) == 1 and gamma.stride(0) == beta.stride(0)
    assert dgamma.stride(0) == dbeta.stride(0)
    dz = torch.empty_like(z)
    L.check(L.lib().vnqa_film_relu_res_bwd_ld(L.ptr(dout), L.ptr(z), L.vptr(gamma), L.vptr(beta), L.ptr(dz), L.vptr(dgamma),
                                              L.vptr(dbeta), N, hp, wp, c, gamma.stride(0), int(film_c), dgamma.stride(0),
                                              L.dtype_id(z.dtype), L.stream()), "vnqa_film_relu_res_bwd_ld")
    return dz


def conv2d_ring(x, wt, bias, H, W, out_padded=None):
    """conv3x3 (weights wt [c_out][9][c_in], K-major) + bias at the outside-ring positions of halo-2 images
    x [n, H+4, W+4, c_in] -> y1 [n * (2(W+2) + 2H), c_out]; implicit GEMM (vnqa_conv2d_ring_fwd).
    out_padded: a ZEROED buffer [n, R+4, c_out] whose four separator rows are never written (see ring_edge_conv)."""
    n, _, _, c_in = x.shape
    c_out = wt.shape[0]
    R = 2 * (W + 2) + 2 * H
    opt = 0
    if out_padded is not None:
        assert out_padded.shape == (n, R + 4, c_out) and out_padded.dtype == x.dtype and out_padded.is_contiguous()
        y1 = out_padded
    else:
        y1 = torch.empty((n * R, c_out), dtype=x.dtype, device=x.device)
    L.check(L.lib().vnqa_conv2d_ring_fwd(L.ptr(x), L.ptr(wt), L.ptr(bias), L.ptr(y1), n, H, W, c_in, c_out,
                                         1 if out_padded is not None else 0, L.dtype_id(x.dtype) | opt, L.stream()),
            "vnqa_conv2d_ring_fwd")
    return y1


RING_EDGE_TAPS = ((6, 7, 8), (0, 1, 2), (2, 5, 8), (0, 3, 6))      # kernel row 2 / row 0 / column 2 / column 0: what the top / bottom / left / right edge sees


def ring_edge_weights(wt):
    """The four 3-tap slices [c_out][3][c_in] of K-major 3x3 weights wt [c_out][9][c_in] that conv2d_ring_edges multiplies with."""
    return [wt[:, list(t), :].contiguous() for t in RING_EDGE_TAPS]


def conv2d_ring_edges(x, wt_edges, bias, H, W, out_padded):
    """conv2d_ring(..., out_padded) as four launches, one per edge, each with the three taps that can see the image (a ring position has
    the other six in the zero halo): K = 3 c_in instead of 9 c_in, the same sums in the same order (vnqa_conv2d_ring_edge_fwd)."""
    n, _, _, c_in = x.shape
    c_out = wt_edges[0].shape[0]
    R = 2 * (W + 2) + 2 * H
    assert out_padded.shape == (n, R + 4, c_out) and out_padded.dtype == x.dtype and out_padded.is_contiguous()
    for e in range(4):
        assert wt_edges[e].shape == (c_out, 3, c_in) and wt_edges[e].is_contiguous() and wt_edges[e].dtype == x.dtype
        be = bias[e] if isinstance(bias, (list, tuple)) else bias          # (per-edge biases: mean-shifted storage, stem._setup_mean_shift)
        L.check(L.lib().vnqa_conv2d_ring_edge_fwd(L.ptr(x), L.ptr(wt_edges[e]), L.ptr(be), L.ptr(out_padded), n, H, W, c_in, c_out, e,
                                                  L.dtype_id(x.dtype), L.stream()), "vnqa_conv2d_ring_edge_fwd")
    return out_padded


def ring_edge_conv(y1p, wt_edge, H, W, edge):
    """One edge product of the border correction as an implicit 1x3 conv along the padded ring rows (vnqa_ring_edge_conv_fwd):
    y1p [n, R+4, cm], wt_edge [co, 3*cm] -> [n * (W | H), co]."""
    n, _, cm = y1p.shape
    co = wt_edge.shape[0]
    ln = W if edge < 2 else H
    opt = 0
    out = torch.empty((n * ln, co), dtype=y1p.dtype, device=y1p.device)
    L.check(L.lib().vnqa_ring_edge_conv_fwd(L.ptr(y1p), L.ptr(wt_edge), L.ptr(out), n, H, W, cm, co, edge,
                                            L.dtype_id(y1p.dtype) | opt, L.stream()), "vnqa_ring_edge_conv_fwd")
    return out


def ring_im2col(x, H, W):
    """x halo-2 padded NHWC [n,H+4,W+4,c] -> [n*(2(W+2)+2H), 9*c]: 3x3 patches around the outside-ring positions."""
    n, _, _, c = x.shape
    R = 2 * (W + 2) + 2 * H
    out = torch.empty((n * R, 9 * c), dtype=x.dtype, device=x.device)
    L.check(L.lib().vnqa_ring_im2col(L.ptr(x), L.ptr(out), n, H, W, c, L.dtype_id(x.dtype), L.stream()), "vnqa_ring_im2col")
    return out


def ring_edge_gather(y1, n, H, W, edge):
    """y1 [n*ring, c] -> [n*(W|H), 3*c] for edge 0/1/2/3 = top/bottom/left/right."""
    c = y1.shape[-1]
    ln = W if edge < 2 else H
    out = torch.empty((n * ln, 3 * c), dtype=y1.dtype, device=y1.device)
    L.check(L.lib().vnqa_ring_edge_gather(L.ptr(y1), L.ptr(out), n, H, W, c, edge, L.dtype_id(y1.dtype), L.stream()),
            "vnqa_ring_edge_gather")
    return out


def ring_edge_gather_all(y1, n, H, W, out=None):
    """All four edge operands in one launch: [4, group_rows, 3*c] with group_rows = n*max(H,W) rounded up to the 256-row
    GEMM tile (edge e's n*(W|H) rows first, the pad rows are never read back)."""
    c = y1.shape[-1]
    group_rows = L.round_up(n * max(H, W), 256)
    if out is None:
        out = torch.empty((4, group_rows, 3 * c), dtype=y1.dtype, device=y1.device)   # pad rows: never read back
    assert out.shape == (4, group_rows, 3 * c) and out.is_contiguous()
    L.check(L.lib().vnqa_ring_edge_gather_all(L.ptr(y1), L.ptr(out), n, H, W, c, group_rows, L.dtype_id(y1.dtype),
                                              L.stream()), "vnqa_ring_edge_gather_all")
    return out


def gemm_nt_grouped(a, b, out=None):
    """out[g] = a[g] @ b[g].T for a [G, Mg, K], b [G, N, K] in one launch (Mg a multiple of 256)."""
    G, Mg, Kd = a.shape
    N = b.shape[1]
    assert b.shape == (G, N, Kd) and a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous()
    if out is None:
        out = torch.empty((G, Mg, N), dtype=a.dtype, device=a.device)
    L.check(L.lib().vnqa_gemm_nt_grouped(L.ptr(a), L.ptr(b), L.ptr(out), G, Mg, N, Kd, out.stride(1),
                                         L.dtype_id(a.dtype), L.stream()), "vnqa_gemm_nt_grouped")
    return out


def ring_assemble(top, bottom, left, right, n, H, W, corner=None):
    """The four edge parts [n * (W | H), c] -> border_sub [n, 2W + 2(H-2), c] (corner pixels: the sum of their two edges, minus
    `corner` [n, 4 * c] — top-left, top-right, bottom-left, bottom-right — when given: vnqa_ring_assemble_corners)."""
    c = top.shape[-1]
    ring = torch.empty((n, 2 * W + 2 * (H - 2), c), dtype=top.dtype, device=top.device)
    if corner is not None:
        assert corner.shape == (n, 4 * c) and corner.dtype == top.dtype and corner.is_contiguous()
    L.check(L.lib().vnqa_ring_assemble_corners(L.ptr(top), L.ptr(bottom), L.ptr(left), L.ptr(right), L.ptr(corner), L.ptr(ring), n, H, W, c,
                                               L.dtype_id(top.dtype), L.stream()), "vnqa_ring_assemble")
    return ring


def border_edge_conv(x, wt5, bias, H, W, edge):
    """One edge of the composed pair's border correction as a 1x5 / 5x1 conv over the image's border row / column
    (vnqa_conv2d_border_edge_fwd): x halo-2 padded NHWC [n, H+4, W+4, c_in], wt5 [c_out, 5, c_in] -> [n * (W | H), c_out]."""
    n, _, _, c_in = x.shape
    c_out = wt5.shape[0]
    assert wt5.shape == (c_out, 5, c_in) and wt5.dtype == x.dtype and wt5.is_contiguous()
    out = torch.empty((n * (W if edge < 2 else H), c_out), dtype=x.dtype, device=x.device)
    L.check(L.lib().vnqa_conv2d_border_edge_fwd(L.ptr(x), L.ptr(wt5), L.ptr(bias), L.ptr(out), n, H, W, c_in, c_out, edge,
                                                L.dtype_id(x.dtype), L.stream()), "vnqa_conv2d_border_edge_fwd")
    return out


def conv2d_c64(x, wt, bias=None, relu=False, pool2=False, post_scale=None, post_shift=None, out=None, shape4=False,
               reserve_cus=0):
    """Persistent direct conv for Cin == 64 (bf16, 3x3): same tensors as conv2d_igemm."""
    N, Hp, Wp, Cin = x.shape
    H, W = Hp - 2, Wp - 2
    c_out, taps, cin_w = wt.shape
    assert Cin == 64 and cin_w == 64 and taps == 9 and L.is_half(x.dtype) and wt.dtype == x.dtype
    Ho, Wo = (H // 2, W // 2) if pool2 else (H, W)
    if out is None:
        out = torch.zeros((N, Ho + 2, Wo + 2, c_out), dtype=x.dtype, device=x.device)
    d = L.ConvDesc(L.BF16, N, H, W, Cin, c_out, out.shape[-1], 9, 1, 1, int(relu), 1 if pool2 else 0, 2 if shape4 else 0, 0, 0,
                   L.conv_reserve_flags(reserve_cus))
    L.check(L.lib().vnqa_conv2d_c64_fwd(ctypes.byref(d), L.ptr(x), L.ptr(wt), L.ptr(bias), L.ptr(post_scale),
                                        L.ptr(post_shift), L.ptr(out), L.stream()), "vnqa_conv2d_c64_fwd")
    return out


def _wreg_desc(x, wt, pool2, relu, out_c, y_halo, reserve_cus=0):
    N, Hp, Wp, Cin = x.shape
    return L.ConvDesc(L.BF16, N, Hp - 2, Wp - 2, Cin, wt.shape[0], out_c, 9, 1, y_halo, int(relu), 1 if pool2 else 0, 0, 0, 0,
                      L.conv_reserve_flags(reserve_cus))


def conv2d_wreg_supported(x, wt, pool2=False, y_halo=1):
    """True when the weights-in-registers direct conv (vnqa_conv2d_wreg_fwd) serves this layer."""
    if not (L.is_half(x.dtype) and wt.dim() == 3 and wt.shape[1] == 9 and wt.shape[2] == x.shape[-1]):
        return False
    d = _wreg_desc(x, wt, pool2, False, wt.shape[0], y_halo)
    return bool(L.lib().vnqa_conv2d_wreg_supported(ctypes.byref(d)))


def conv2d_wreg(x, wt, bias=None, relu=False, pool2=False, post_scale=None, post_shift=None, out=None, y_halo=1,
                reserve_cus=0):
    """Weights-stationary-in-registers persistent direct 3x3 conv (csrc/conv_wreg.hip): same tensors as conv2d_igemm
    (x padded NHWC halo 1, wt [c_out, 9, c_in] K-major), for conv1_2 / conv2_1 / conv2_2 shaped layers."""
    N, Hp, Wp, Cin = x.shape
    H, W = Hp - 2, Wp - 2
    c_out = wt.shape[0]
    assert wt.shape[1] == 9 and wt.shape[2] == Cin and L.is_half(x.dtype) and wt.dtype == x.dtype
    Ho, Wo = (H // 2, W // 2) if pool2 else (H, W)
    if out is None:
        out = torch.zeros((N, Ho + 2 * y_halo, Wo + 2 * y_halo, c_out), dtype=x.dtype, device=x.device)
    assert out.shape[:3] == (N, Ho + 2 * y_halo, Wo + 2 * y_halo)
    d = _wreg_desc(x, wt, pool2, relu, out.shape[-1], y_halo, reserve_cus)
    L.check(L.lib().vnqa_conv2d_wreg_fwd(ctypes.byref(d), L.ptr(x), L.ptr(wt), L.ptr(bias), L.ptr(post_scale),
                                         L.ptr(post_shift), L.ptr(out), L.stream()), "vnqa_conv2d_wreg_fwd")
    return out


def layernorm_fwd(x, gamma, beta, eps, rows=None):
    """LayerNorm over the last dimension of fp32 rows (optionally gathered: row r of the result normalises x[rows[r]]).
    Returns (y, mean, rstd)."""
    n = x.shape[-1]
    n_rows = rows.numel() if rows is not None else x.numel() // n
    y = torch.empty((n_rows, n), dtype=torch.float32, device=x.device)
    mean = torch.empty((n_rows,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((n_rows,), dtype=torch.float32, device=x.device)
    L.check(L.lib().vnqa_layernorm_fwd(L.ptr(_f32c(x)), L.ptr(rows), L.ptr(_f32c(gamma)), L.ptr(_f32c(beta)), L.ptr(y), L.ptr(mean),
                                       L.ptr(rstd), n_rows, n, float(eps), L.stream()), "vnqa_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, rows=None, dgamma=None, dbeta=None, accumulate=False, need_dx=True):
    """Returns (dx [n_rows, n] or None, dgamma, dbeta); dgamma / dbeta are written (or accumulated) in place when given."""
    n_rows, n = dy.shape
    dx = torch.empty((n_rows, n), dtype=torch.float32, device=dy.device) if need_dx else None
    if dgamma is None:
        dgamma = torch.empty((n,), dtype=torch.float32, device=dy.device)
        dbeta = torch.empty((n,), dtype=torch.float32, device=dy.device)
        accumulate = False
    L.check(L.lib().vnqa_layernorm_bwd(L.ptr(_f32c(dy)), L.ptr(_f32c(x)), L.ptr(rows), L.ptr(mean), L.ptr(rstd), L.ptr(_f32c(gamma)),
                                       L.ptr(dx), L.ptr(dgamma), L.ptr(dbeta), n_rows, n, 1 if accumulate else 0, L.stream()),
            "vnqa_layernorm_bwd")
    return dx, dgamma, dbeta


def scatter_add_rows(dst, rows, src):
    """dst[rows[r]] += src[r] (unique rows; fp32 2-D, contiguous)."""
    n_rows, n = src.shape
    assert dst.is_contiguous() and dst.dtype == torch.float32 and dst.shape[-1] == n
    L.check(L.lib().vnqa_scatter_add_rows(L.ptr(dst), L.ptr(rows), L.ptr(_f32c(src)), n_rows, n, L.stream()), "vnqa_scatter_add_rows")
    return dst


def hop_fwd(hv, hs, base_row, qlen, w, bias, lmax):
    """one hop of the multi-hop FiLM generator for every image (vnqa_hop_fwd): returns (hv_out [n_img, H], coefs [n_img, lmax])."""
    n_img, H = hv.shape
    out = torch.empty((n_img, H), dtype=torch.float32, device=hv.device)
    coefs = torch.empty((n_img, lmax), dtype=torch.float32, device=hv.device)
    L.check(L.lib().vnqa_hop_fwd(L.ptr(_f32c(hv)), L.ptr(_f32c(hs)), L.ptr(base_row), L.ptr(qlen), L.ptr(_f32c(w.reshape(-1))),
                                 L.ptr(_f32c(bias.reshape(-1))), L.ptr(out), L.ptr(coefs), n_img, lmax, H, L.stream()), "vnqa_hop_fwd")
    return out, coefs


def hop_bwd(dout, hv, hs, base_row, qlen, w, coefs, dhs):
    """backward of hop_fwd: returns (d hv [n_img, H], per-image d w rows [n_img, H]); d hs is ACCUMULATED into `dhs`."""
    n_img, H = hv.shape
    dhv = torch.empty((n_img, H), dtype=torch.float32, device=hv.device)
    dw_img = torch.empty((n_img, H), dtype=torch.float32, device=hv.device)
    L.check(L.lib().vnqa_hop_bwd(L.ptr(_f32c(dout)), L.ptr(_f32c(hv)), L.ptr(_f32c(hs)), L.ptr(base_row), L.ptr(qlen),
                                 L.ptr(_f32c(w.reshape(-1))), L.ptr(coefs), L.ptr(dhv), L.ptr(dhs), L.ptr(dw_img), n_img,
                                 coefs.shape[1], H, L.stream()), "vnqa_hop_bwd")
    return dhv, dw_img


def frame_max_fwd(maps, frame_off_i32, batch, n_frames, tail):
    """max over a sample's frames of the packed relu'd tail maps [n_img, h+2, w+2, c_pad] -> (pooled fp32 [batch, tail*h*w] in
    NCHW-flattened order, argmax int32) — vnqa_frame_max_fwd."""
    n_img, hp, wp, c_pad = maps.shape
    h, w = hp - 2, wp - 2
    pooled = torch.empty((batch, tail * h * w), dtype=torch.float32, device=maps.device)
    argmax = torch.empty((batch, tail * h * w), dtype=torch.int32, device=maps.device)
    L.check(L.lib().vnqa_frame_max_fwd(L.ptr(maps.contiguous()), L.ptr(frame_off_i32), L.ptr(pooled), L.ptr(argmax), batch,
                                       n_frames, h, w, c_pad, tail, L.dtype_id(maps.dtype), L.stream()), "vnqa_frame_max_fwd")
    return pooled, argmax


def frame_max_bwd(dpooled, argmax, sample_of_i32, shape, dtype, tail, scale=1.0):
    """gradient of frame_max_fwd w.r.t. the packed maps: the whole padded NHWC tensor `shape` = [n_img, h+2, w+2, c_pad]."""
    n_img, hp, wp, c_pad = shape
    dmaps = torch.empty(shape, dtype=dtype, device=dpooled.device)
    L.check(L.lib().vnqa_frame_max_bwd(L.ptr(dpooled.contiguous()), L.ptr(argmax), L.ptr(sample_of_i32), L.ptr(dmaps), n_img,
                                       hp - 2, wp - 2, c_pad, tail, float(scale), L.dtype_id(dtype), L.stream()),
            "vnqa_frame_max_bwd")
    return dmaps


_PIXEL_LUT = {}


def pixel_lut(device):
    """lut[k] = float32(k / 255.0) with the division in float64: exactly the value a raw pixel k has after the reference's
    `clip / 255.0` on a float64 tensor (eval/dataset.py:91) and the training loop's `.float()` (eval/q_and_v_eval.py:92)."""
    key = str(device)
    t = _PIXEL_LUT.get(key)
    if t is None:
        t = _PIXEL_LUT[key] = (torch.arange(256, dtype=torch.float64) / 255.0).float().to(device)
    return t


def expand_u8_clip(clip):
    """uint8 clip -> the fp32 clip the reference's loader would have produced (device-side table lookup)."""
    return pixel_lut(clip.device)[clip.long()]


def clip_to_nhwc4(clip, img_of, n_img, out=None, shift=None):
    """clip fp32 — or RAW uint8 pixels valued k / 255 — [B,3,H,W,T] (frames last) -> image list bf16 [n_img,H+4,W+4,4]
    (halo 2 and channel 3 zero).  shift (device fp32 [3]): the list holds pixel - shift[c] (mean-shifted storage; the caller keeps
    -shift[c] in the halo of `out`)."""
    B, C, H, W, T = clip.shape
    assert C == 3 and clip.dtype in (torch.float32, torch.uint8)
    if out is None:
        out = torch.zeros((n_img, H + 4, W + 4, 4), dtype=L.half_dtype(), device=clip.device)
    if shift is not None:
        assert shift.dtype == torch.float32 and shift.numel() >= 3 and shift.is_cuda
        lut = pixel_lut(clip.device) if clip.dtype == torch.uint8 else None
        L.check(L.lib().vnqa_clip_to_nhwc4_shifted(L.ptr(clip.contiguous()), L.ptr(lut), L.ptr(shift), L.ptr(img_of), L.ptr(out),
                                                   B, T, H, W, L.stream()), "vnqa_clip_to_nhwc4_shifted")
        return out
    if clip.dtype == torch.uint8:
        L.check(L.lib().vnqa_clip_u8_to_nhwc4(L.ptr(clip.contiguous()), L.ptr(pixel_lut(clip.device)), L.ptr(img_of), L.ptr(out),
                                              B, T, H, W, L.stream()), "vnqa_clip_u8_to_nhwc4")
        return out
    L.check(L.lib().vnqa_clip_to_nhwc4(L.ptr(clip.contiguous()), L.ptr(img_of), L.ptr(out), B, T, H, W, L.stream()),
            "vnqa_clip_to_nhwc4")
    return out


def conv_first_c64(img4, w1, b1, wt, bias=None, relu=False, pool2=False, post_scale=None, post_shift=None, out=None,
                   reserve_cus=0, sched=None, mid_shift=None):
    """conv(3->64)+ReLU fused into the following C_in = 64 conv (see vnqa_conv_first_c64_fwd).  img4 from clip_to_nhwc4;
    w1/b1 the first conv's fp32 OIHW weights and bias; wt/bias/... as conv2d_c64.  sched: int32 [2] device tensor, zero on first
    use — the dynamic tile schedule (vnqa_conv_first_c64_fwd_sched); None = static stride."""
    N, Hp4, Wp4, c4 = img4.shape
    H, W = Hp4 - 4, Wp4 - 4
    c_out, taps, cin_w = wt.shape
    assert c4 == 4 and cin_w == 64 and taps == 9 and L.is_half(img4.dtype) and wt.dtype == img4.dtype
    Ho, Wo = (H // 2, W // 2) if pool2 else (H, W)
    if out is None:
        out = torch.zeros((N, Ho + 2, Wo + 2, c_out), dtype=img4.dtype, device=img4.device)
    flags = L.conv_reserve_flags(reserve_cus)
    b1 = b1.detach().float().contiguous()
    if mid_shift is not None:      # VNQA_CONV_FIRST_MID_SHIFT: b1 = [bias | shift of the first conv's stored output] (callers cache the pair)
        if b1.numel() == 64:
            b1 = torch.cat([b1, mid_shift.detach().float().reshape(-1)[:64]]).contiguous()
        assert b1.numel() == 128
        flags |= L.CONV_FIRST_MID_SHIFT
    d = L.ConvDesc(L.BF16, N, H, W, 64, c_out, out.shape[-1], 9, 1, 1, int(relu), 1 if pool2 else 0, 0, 0, 0, flags)
    L.check(L.lib().vnqa_conv_first_c64_fwd_sched(ctypes.byref(d), L.ptr(img4), L.ptr(w1.detach().float().contiguous()),
                                                  L.ptr(b1), L.ptr(wt), L.ptr(bias),
                                                  L.ptr(post_scale), L.ptr(post_shift), L.ptr(out), L.ptr(sched), L.stream()),
            "vnqa_conv_first_c64_fwd")
    return out


def pack_fc_weight(w, C, h, wd, c_pad, rows_pad, dtype, want_t=True):
    """nn.Linear weight over an NCHW-flattened map [rows, C*h*wd] fp32 -> (nat [rows_pad, (h+2)(wd+2)*c_pad],
    nat_t [(h+2)(wd+2)*c_pad, rows_pad] or None) in `dtype`."""
    rows = w.shape[0]
    kn = (h + 2) * (wd + 2) * c_pad
    w = w.detach().float().contiguous()
    nat = torch.empty((rows_pad, kn), dtype=dtype, device=w.device)
    nat_t = torch.empty((kn, rows_pad), dtype=dtype, device=w.device) if want_t else None
    L.check(L.lib().vnqa_pack_fc_weight(L.ptr(w), rows, C, h, wd, rows_pad, c_pad, L.dtype_id(dtype), L.ptr(nat),
                                        L.ptr(nat_t), L.stream()), "vnqa_pack_fc_weight")
    return nat, nat_t


def fc_dx_supported(m, rows_pad, kn, dtype):
    """vnqa_fc_dx's shapes: 16-bit storage, contraction 128, a whole number of 128-column slabs."""
    return L.is_half(dtype) and rows_pad == 128 and m > 0 and kn % 128 == 0


def fc_dx(dout, nat):
    """dx [m, kn] = dout [m, 128] @ nat [128, kn] (16-bit): the Linear's input gradient from its forward operand."""
    m, r = dout.shape
    kn = nat.shape[1]
    assert nat.shape[0] == r and dout.dtype == nat.dtype and dout.is_contiguous() and nat.is_contiguous()
    dx = torch.empty((m, kn), dtype=dout.dtype, device=dout.device)
    L.check(L.lib().vnqa_fc_dx(L.ptr(dout), L.ptr(nat), L.ptr(dx), m, r, kn, L.dtype_id(dout.dtype), L.stream()), "vnqa_fc_dx")
    return dx


def unpack_fc_wgrad(dw_nat, rows, C, h, wd, c_pad, out=None, alpha=1.0):
    """fp32 gradient of the native-layout weight [rows_pad, (h+2)(wd+2)*c_pad] -> [rows, C*h*wd]."""
    dw = out if out is not None else torch.empty((rows, C * h * wd), dtype=torch.float32, device=dw_nat.device)
    assert dw.shape == (rows, C * h * wd) and dw.is_contiguous() and dw.dtype == torch.float32
    L.check(L.lib().vnqa_unpack_fc_wgrad_dev(L.ptr(dw_nat), rows, C, h, wd, c_pad, L.ptr(dw), float(alpha),
                                             None, L.stream()), "vnqa_unpack_fc_wgrad")
    return dw


def conv_first(clip, w, bias, img_of, n_img, dtype, out=None):
    """clip fp32 [B,3,H,W,T] -> padded NHWC [n_img,H+2,W+2,64] (conv1_1 + ReLU)."""
    B, C, H, W, T = clip.shape
    assert C == 3 and clip.dtype == torch.float32
    c_out = w.shape[0]
    if out is None:
        out = torch.zeros((n_img, H + 2, W + 2, c_out), dtype=dtype, device=clip.device)
    L.check(L.lib().vnqa_conv_first_fwd(L.ptr(clip), L.ptr(w.detach().float().contiguous()),
                                        L.ptr(bias.detach().float().contiguous()), L.ptr(img_of), L.ptr(out),
                                        B, T, H, W, c_out, L.dtype_id(dtype), L.stream()), "vnqa_conv_first_fwd")
    return out


def feat_to_nhwc(v, img_of, n_img, dtype, c_pad=None):
    """v fp32 [B,C,h,w,T] -> padded NHWC [n_img,h+2,w+2,c_pad]; frame (b,t) -> image img_of[b*T+t]."""
    B, C, h, w, T = v.shape
    c_pad = c_pad or L.round_up(C, 64)
    out = torch.zeros((n_img, h + 2, w + 2, c_pad), dtype=dtype, device=v.device)
    L.check(L.lib().vnqa_feat_to_nhwc(L.ptr(v.float().contiguous()), L.ptr(img_of), L.ptr(out), B, C, h, w, T,
                                      c_pad, L.dtype_id(dtype), L.stream()), "vnqa_feat_to_nhwc")
    return out


def nchw_to_nhwc(x, dtype, c_pad=None):
    N, C, h, w = x.shape
    c_pad = c_pad or L.round_up(C, 64)
    out = torch.zeros((N, h + 2, w + 2, c_pad), dtype=dtype, device=x.device)
    L.check(L.lib().vnqa_nchw_to_nhwc(L.ptr(x.float().contiguous()), L.ptr(out), N, C, h, w, c_pad,
                                      L.dtype_id(dtype), L.stream()), "vnqa_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x, C, halo=1):
    N, Hp, Wp, c_pad = x.shape
    h, w = Hp - 2 * halo, Wp - 2 * halo
    out = torch.empty((N, C, h, w), dtype=torch.float32, device=x.device)
    L.check(L.lib().vnqa_nhwc_to_nchw(L.ptr(x), L.ptr(out), N, C, h, w, c_pad, halo, L.dtype_id(x.dtype),
                                      L.stream()), "vnqa_nhwc_to_nchw")
    return out


_WS = {}


def workspace(nbytes, device):
    """Grow-only fp32 scratch buffer per device AND stream (caller-owned workspace of the C ABI): the frozen stem
    runs its split-K GEMMs on a side stream while the trunk uses its own on the main stream."""
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    n = (int(nbytes) + 3) // 4
    buf = _WS.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.empty(max(n, 1), dtype=torch.float32, device=device)
        _WS[key] = buf
    return buf


WGRAD_EIGHT_WAVES = False      # A/B switch: the first 16-bit form of the weight-gradient kernel for every call


def conv2d_wgrad(x, dy, taps, want_bias=True, dbias_out=None, x_segs=1, eight_waves=False):
    """x, dy: padded NHWC (halo 1, same N/H/W). Returns (dwt fp32 [Cout][taps][Cin], dbias fp32 [Cout]).
    x_segs = 2 / 3: x is a [hi | lo] / [hi | lo | hi] tensor (x_segs Cin physical channels); its first segment — the plain 16-bit
    value — is contracted in place.  eight_waves: the first 16-bit form of the kernel (VNQA_WGRAD_EIGHT_WAVES)."""
    N, Hp, Wp, Cin = x.shape
    Cout = dy.shape[-1]
    h, w = Hp - 2, Wp - 2
    if x_segs > 1:
        assert x_segs in (2, 3) and L.is_half(x.dtype) and Cin % x_segs == 0
        Cin //= x_segs
    assert dy.shape[:3] == x.shape[:3] and dy.dtype == x.dtype
    ws = workspace(L.lib().vnqa_conv2d_wgrad_workspace(N, h, w, Cin, Cout, taps), x.device)
    dwt = torch.empty((Cout, taps, Cin), dtype=torch.float32, device=x.device)
    dbias = None
    if want_bias:
        dbias = dbias_out if (dbias_out is not None and dbias_out.numel() == Cout) else \
            torch.empty((Cout,), dtype=torch.float32, device=x.device)
    L.check(L.lib().vnqa_conv2d_wgrad(L.ptr(x), L.ptr(dy), L.ptr(dwt), L.ptr(dbias), L.ptr(ws), N, h, w, Cin, Cout,
                                      taps, L.dtype_id(x.dtype) | {1: 0, 2: L.WGRAD_X_PAIR, 3: L.WGRAD_X_TRIPLE}[x_segs] |
                                      (L.WGRAD_EIGHT_WAVES if (eight_waves or WGRAD_EIGHT_WAVES) and L.is_half(x.dtype) else 0), L.stream()),
            "vnqa_conv2d_wgrad")
    return dwt, dbias


def unpack_conv_wgrad(dwt, c_out, c_in, out=None, alpha=1.0):
    """fp32 [c_out_pad][taps][c_in_pad] -> OIHW fp32 [c_out][c_in][k][k]."""
    c_out_pad, taps, c_in_pad = dwt.shape
    shape = {1: (1, 1), 9: (3, 3), 27: (3, 3, 3)}[taps]
    if out is None:
        out = torch.empty((c_out, c_in) + shape, dtype=torch.float32, device=dwt.device)
    assert out.shape == (c_out, c_in) + shape and out.is_contiguous() and out.dtype == torch.float32
    L.check(L.lib().vnqa_unpack_conv_wgrad_dev(L.ptr(dwt), c_out, c_in, taps, c_out_pad, c_in_pad, L.ptr(out), float(alpha),
                                               None, L.stream()), "vnqa_unpack_conv_wgrad")
    return out


def gemm_nt(a, b, bias=None, relu=False, out=None, split_k=True, split_weights=False):
    """out[m][n] = act(sum_k a[m][k] b[n][k] + bias[n]); a [M,K], b [N,K] (same dtype), out dtype = a.dtype.
    split_k=False: one pass over K in a fixed order (result independent of M's tiling).
    split_weights (precision 'fp16h'): b is the fp32 operand — a . b_hi + a . b_lo, a read twice along K (VNQA_GEMM_X_WRAP2)."""
    M, Kd = a.shape
    N = b.shape[0]
    opt, kk = 0, Kd
    if split_weights:
        assert L.is_half(a.dtype) and b.dtype == torch.float32 and Kd % 64 == 0 and a.is_contiguous()
        b, opt, kk = split_weight2(b), L.GEMM_X_WRAP2, 2 * Kd
    assert b.shape[1] == kk and a.dtype == b.dtype
    if out is None:
        out = torch.empty((M, N), dtype=a.dtype, device=a.device)
    did = L.dtype_id(a.dtype)
    ws_bytes = L.lib().vnqa_gemm_nt_workspace(M, N, kk, did) if split_k else 0
    ws = workspace(ws_bytes, a.device) if ws_bytes > 0 else None
    L.check(L.lib().vnqa_gemm_nt(L.ptr(a), L.ptr(b), L.ptr(bias), L.ptr(out), L.ptr(ws), M, N, kk, out.stride(0),
                                 1 if relu else 0, did | opt, L.stream()), "vnqa_gemm_nt")
    return out


def gemm_tn(a, b, out=None):
    """out[m][n] = sum_k a[k][m] b[k][n]; a [K,M], b [K,N] (same dtype) -> fp32 [M,N]."""
    Kd, M = a.shape
    N = b.shape[1]
    assert b.shape[0] == Kd and a.dtype == b.dtype
    did = L.dtype_id(a.dtype)
    ws = workspace(L.lib().vnqa_gemm_tn_workspace(M, N, Kd, did), a.device)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    assert out.shape == (M, N) and out.is_contiguous() and out.dtype == torch.float32
    L.check(L.lib().vnqa_gemm_tn(L.ptr(a), L.ptr(b), L.ptr(out), L.ptr(ws), M, N, Kd, did, L.stream()),
            "vnqa_gemm_tn")
    return out


def clip_adam_step(p, g, m, v, partial, step, lr, clip=1.0, beta1=0.9, beta2=0.999, eps=1e-8, overflow_count=None):
    """Fused clip_grad_norm + Adam + zero_grad on flat fp32 buffers (in place).  overflow_count (device int32 [1], loss-scaled
    fp16 training): a non-finite gradient norm skips the update on the device and increments it."""
    n = p.numel()
    nb = L.lib().vnqa_l2norm_blocks(n)
    assert partial.numel() >= nb
    L.check(L.lib().vnqa_l2norm_partial(L.ptr(g), n, L.ptr(partial), L.stream()), "vnqa_l2norm_partial")
    L.check(L.lib().vnqa_clip_adam(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), n, L.ptr(partial), nb, clip, lr,
                                   beta1, beta2, eps, step, L.ptr(overflow_count), L.stream()), "vnqa_clip_adam")


def lstm_seq_fwd(xg, w_hh, q_lens_i32, h0, c0, n_rep, S):
    """Persistent LSTM forward.  xg [B,Lq,4H] fp32; returns hs [B,S,H], gates [B,S,5H], hN, cN."""
    B, Lq, H4 = xg.shape
    H = H4 // 4
    dev = xg.device
    hs = torch.empty((B, S, H), dtype=torch.float32, device=dev)         # the kernel zero-fills the rows past each chain
    gates = torch.empty((B, S, 5 * H), dtype=torch.float32, device=dev)  # rows past a chain's end are never read
    hN = torch.empty((B, H), dtype=torch.float32, device=dev)
    cN = torch.empty((B, H), dtype=torch.float32, device=dev)
    L.check(L.lib().vnqa_lstm_seq_fwd(L.ptr(xg), L.ptr(w_hh), L.ptr(q_lens_i32), L.ptr(h0), L.ptr(c0), L.ptr(hs),
                                      L.ptr(gates), L.ptr(hN), L.ptr(cN), B, Lq, H, S, n_rep, L.stream()),
            "vnqa_lstm_seq_fwd")
    return hs, gates, hN, cN


def lstm_seq_bwd(w_hh, q_lens_i32, c0, gates, dhs, dhN, dcN, n_rep):
    """Persistent BPTT.  Returns dgates [B,S,4H], dh0, dc0."""
    B, S, H5 = gates.shape
    H = H5 // 5
    dev = gates.device
    dgates = torch.empty((B, S, 4 * H), dtype=torch.float32, device=dev)  # tail rows zero-filled by the kernel
    dh0 = torch.empty((B, H), dtype=torch.float32, device=dev)
    dc0 = torch.empty((B, H), dtype=torch.float32, device=dev)
    L.check(L.lib().vnqa_lstm_seq_bwd(L.ptr(w_hh), L.ptr(q_lens_i32), L.ptr(c0), L.ptr(gates), L.ptr(dhs),
                                      L.ptr(dhN), L.ptr(dcN), L.ptr(dgates), L.ptr(dh0), L.ptr(dc0), B, H, S,
                                      n_rep, L.stream()), "vnqa_lstm_seq_bwd")
    return dgates, dh0, dc0


def lstm_wide_fwd(xg, w_hh, batch_sizes, reverse=False, h0=None, c0=None):
    """Step-wise packed LSTM for wide hidden states.  xg fp32 [T,B,4H] time-major, batch_sizes a host int
    sequence (PackedSequence.batch_sizes).  Returns hs, cs [T,B,H] and gates [T,B,4H] (zero on inactive rows)."""
    T, B, H4 = xg.shape
    H = H4 // 4
    dev = xg.device
    bs = (ctypes.c_int32 * T)(*[int(v) for v in batch_sizes])
    hs = torch.zeros((T, B, H), dtype=torch.float32, device=dev)
    cs = torch.zeros((T, B, H), dtype=torch.float32, device=dev)
    gates = torch.zeros((T, B, H4), dtype=torch.float32, device=dev)
    L.check(L.lib().vnqa_lstm_wide_fwd(L.ptr(xg), L.ptr(w_hh), L.ptr(h0), L.ptr(c0), bs, L.ptr(hs), L.ptr(cs),
                                       L.ptr(gates), T, B, H, 1 if reverse else 0, L.stream()), "vnqa_lstm_wide_fwd")
    return hs, cs, gates


def lstm_wide_bwd(w_hh_t, batch_sizes, gates, cs, dhs, reverse=False, c0=None):
    """BPTT of lstm_wide_fwd.  Returns dgates [T,B,4H] (= d xg)."""
    T, B, H = cs.shape
    dev = cs.device
    bs = (ctypes.c_int32 * T)(*[int(v) for v in batch_sizes])
    dgates = torch.zeros((T, B, 4 * H), dtype=torch.float32, device=dev)
    dc = torch.zeros((B, H), dtype=torch.float32, device=dev)
    L.check(L.lib().vnqa_lstm_wide_bwd(L.ptr(w_hh_t), L.ptr(c0), bs, L.ptr(gates), L.ptr(cs), L.ptr(dhs),
                                       L.ptr(dgates), L.ptr(dc), T, B, H, 1 if reverse else 0, L.stream()),
            "vnqa_lstm_wide_bwd")
    return dgates


def sgemm_batch(problems):
    """vnqa_sgemm_batch: up to 4 independent fp32 products in one launch.  Each problem is a dict with tensors a, b, c and
    optional bias / addend / out2 / out2_col / out2_mul, the element strides a_rs, a_cs, b_rs, b_cs and ldc, m, n, k, relu,
    accumulate (the arguments of vnqa_sgemm / vnqa_sgemm2)."""
    arr = (L.SgemmProblem * len(problems))()
    for q, d in zip(arr, problems):
        for n in ("a", "b", "c", "bias", "addend", "out2", "out2_col", "out2_mul"):
            setattr(q, n, L.ptr(d.get(n)))
        for n in ("a_rs", "a_cs", "b_rs", "b_cs", "ldc", "m", "n", "k"):
            setattr(q, n, int(d[n]))
        q.relu, q.accumulate = int(d.get("relu", 0)), int(d.get("accumulate", 0))
    L.check(L.lib().vnqa_sgemm_batch(ctypes.cast(arr, ctypes.c_void_p), len(problems), L.stream()), "vnqa_sgemm_batch")


def lstm_wide_bidir_fwd(xg_f, xg_r, w_hh_f, w_hh_r, batch_sizes):
    """Both directions of a bidirectional packed LSTM from zero states, one launch per chain position
    (vnqa_lstm_wide_bidir_fwd).  Returns ((hs_f, hs_r), (cs_f, cs_r), (gates_f, gates_r)); bit-identical to two
    lstm_wide_fwd calls."""
    T, B, H4 = xg_f.shape
    H = H4 // 4
    dev = xg_f.device
    assert xg_r.shape == xg_f.shape and w_hh_f.shape == w_hh_r.shape == (H4, H)
    bs = (ctypes.c_int32 * T)(*[int(v) for v in batch_sizes])
    hs = torch.zeros((2, T, B, H), dtype=torch.float32, device=dev)
    cs = torch.zeros((2, T, B, H), dtype=torch.float32, device=dev)
    gates = torch.zeros((2, T, B, H4), dtype=torch.float32, device=dev)
    L.check(L.lib().vnqa_lstm_wide_bidir_fwd(L.ptr(xg_f), L.ptr(xg_r), L.ptr(w_hh_f), L.ptr(w_hh_r), bs, L.ptr(hs[0]),
                                             L.ptr(hs[1]), L.ptr(cs[0]), L.ptr(cs[1]), L.ptr(gates[0]), L.ptr(gates[1]),
                                             T, B, H, L.stream()), "vnqa_lstm_wide_bidir_fwd")
    return (hs[0], hs[1]), (cs[0], cs[1]), (gates[0], gates[1])


def lstm_wide_bidir_bwd(w_hh_t_f, w_hh_t_r, batch_sizes, gates_f, gates_r, cs_f, cs_r, dhs_f, dhs_r):
    """BPTT of lstm_wide_bidir_fwd.  Returns (dgates_f, dgates_r), each [T,B,4H] (= d xg of its direction)."""
    T, B, H = cs_f.shape
    dev = cs_f.device
    bs = (ctypes.c_int32 * T)(*[int(v) for v in batch_sizes])
    dgates = torch.zeros((2, T, B, 4 * H), dtype=torch.float32, device=dev)
    dc = torch.zeros((2, B, H), dtype=torch.float32, device=dev)
    L.check(L.lib().vnqa_lstm_wide_bidir_bwd(L.ptr(w_hh_t_f), L.ptr(w_hh_t_r), bs, L.ptr(gates_f), L.ptr(gates_r), L.ptr(cs_f),
                                             L.ptr(cs_r), L.ptr(dhs_f), L.ptr(dhs_r), L.ptr(dgates[0]), L.ptr(dgates[1]),
                                             L.ptr(dc[0]), L.ptr(dc[1]), T, B, H, L.stream()), "vnqa_lstm_wide_bidir_bwd")
    return dgates[0], dgates[1]


def mac_read_fwd(know, pre, u, v, bias, n, s, c):
    """Fused ReadUnit attention.  know/pre [n*s, ld] (compute dtype), u/v fp32 [n,c] -> p [n,s], read [n,c] fp32."""
    ld = know.shape[-1]
    p = torch.empty((n, s), dtype=torch.float32, device=know.device)
    read = torch.empty((n, c), dtype=torch.float32, device=know.device)
    L.check(L.lib().vnqa_mac_read_fwd(L.ptr(know), L.ptr(pre), L.ptr(u), L.ptr(v), L.ptr(bias), L.ptr(p), L.ptr(read),
                                      n, s, c, ld, L.dtype_id(know.dtype), L.stream()), "vnqa_mac_read_fwd")
    return p, read


def mac_read_bwd(know, pre, p, dread, n, s, c):
    ld = know.shape[-1]
    dscore = torch.empty((n, s), dtype=torch.float32, device=know.device)
    du = torch.empty((n, c), dtype=torch.float32, device=know.device)
    dv = torch.empty((n, c), dtype=torch.float32, device=know.device) if pre is not None else None
    L.check(L.lib().vnqa_mac_read_bwd(L.ptr(know), L.ptr(pre), L.ptr(p), L.ptr(dread), L.ptr(dscore), L.ptr(du),
                                      L.ptr(dv), n, s, c, ld, L.dtype_id(know.dtype), L.stream()), "vnqa_mac_read_bwd")
    return dscore, du, dv


def mac_read_accum(dscore, p, u, v, dread, n, s, c, ld, dtype):
    """Stacked per-step factors ([k,n,s] / [k,n,c] fp32) -> dknow, dpre [n*s, ld] in `dtype`."""
    k = dscore.shape[0]
    dknow = torch.empty((n * s, ld), dtype=dtype, device=dscore.device)
    dpre = torch.empty((n * s, ld), dtype=dtype, device=dscore.device) if v is not None else None
    L.check(L.lib().vnqa_mac_read_accum(L.ptr(dscore), L.ptr(p), L.ptr(u), L.ptr(v), L.ptr(dread), L.ptr(dknow),
                                        L.ptr(dpre), k, n, s, c, ld, L.dtype_id(dtype), L.stream()), "vnqa_mac_read_accum")
    return dknow, dpre


def frame_bn_stats(x, frame_off_i32, n_frames):
    N, hp, wp, c = x.shape
    mean = torch.empty((n_frames, c), dtype=torch.float32, device=x.device)
    var = torch.empty((n_frames, c), dtype=torch.float32, device=x.device)
    L.check(L.lib().vnqa_frame_bn_stats(L.ptr(x), L.ptr(frame_off_i32), L.ptr(mean), L.ptr(var), n_frames, hp, wp, c,
                                        L.dtype_id(x.dtype), L.stream()), "vnqa_frame_bn_stats")
    return mean, var


def frame_bn_stats_split(x_hi, x_lo, frame_off_i32, n_frames):
    """frame_bn_stats of x_hi + x_lo (conv2d_igemm_split_out's pair)."""
    N, hp, wp, c = x_hi.shape
    assert x_lo.shape == x_hi.shape and x_lo.dtype == x_hi.dtype and L.is_half(x_hi.dtype)
    mean = torch.empty((n_frames, c), dtype=torch.float32, device=x_hi.device)
    var = torch.empty((n_frames, c), dtype=torch.float32, device=x_hi.device)
    L.check(L.lib().vnqa_frame_bn_stats_split(L.ptr(x_hi), L.ptr(x_lo), L.ptr(frame_off_i32), L.ptr(mean), L.ptr(var), n_frames, hp, wp, c,
                                              L.stream()), "vnqa_frame_bn_stats_split")
    return mean, var


def frame_bn_apply_split(x_hi, x_lo, frame_of_i32, mean, rstd, gamma, beta):
    N, hp, wp, c = x_hi.shape
    assert x_lo.shape == x_hi.shape and x_lo.dtype == x_hi.dtype and L.is_half(x_hi.dtype)
    y = torch.empty((N, hp, wp, c), dtype=x_hi.dtype, device=x_hi.device)
    L.check(L.lib().vnqa_frame_bn_apply_split(L.ptr(x_hi), L.ptr(x_lo), L.ptr(frame_of_i32), L.ptr(mean), L.ptr(rstd), L.ptr(gamma),
                                              L.ptr(beta), L.ptr(y), N, hp, wp, c, L.stream()), "vnqa_frame_bn_apply_split")
    return y


def frame_bn_apply(x, frame_of_i32, mean, rstd, gamma, beta):
    N, hp, wp, c = x.shape
    y = torch.empty_like(x)
    L.check(L.lib().vnqa_frame_bn_apply(L.ptr(x), L.ptr(frame_of_i32), L.ptr(mean), L.ptr(rstd), L.ptr(gamma),
                                        L.ptr(beta), L.ptr(y), N, hp, wp, c, L.dtype_id(x.dtype), L.stream()),
            "vnqa_frame_bn_apply")
    return y


def frame_bn_bwd(dy, x, frame_of_i32, frame_off_i32, mean, rstd, gamma, n_frames, relu_mask):
    N, hp, wp, c = x.shape
    s1 = torch.empty((n_frames, c), dtype=torch.float32, device=x.device)
    s2 = torch.empty((n_frames, c), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    L.check(L.lib().vnqa_frame_bn_bwd(L.ptr(dy), L.ptr(x), L.ptr(frame_of_i32), L.ptr(frame_off_i32), L.ptr(mean),
                                      L.ptr(rstd), L.ptr(gamma), L.ptr(s1), L.ptr(s2), L.ptr(dx), N, n_frames, hp, wp,
                                      c, 1 if relu_mask else 0, L.dtype_id(x.dtype), L.stream()), "vnqa_frame_bn_bwd")
    return dx, s1, s2


def film_relu_res_fwd(z, res, gamma, beta):
    N, hp, wp, c = z.shape
    out = torch.empty_like(z)
    L.check(L.lib().vnqa_film_relu_res_fwd(L.ptr(z), L.ptr(res), L.ptr(gamma), L.ptr(beta), L.ptr(out), N, hp, wp, c,
                                           L.dtype_id(z.dtype), L.stream()), "vnqa_film_relu_res_fwd")
    return out


def film_relu_res_bwd(dout, z, gamma, beta):
    N, hp, wp, c = z.shape
    dz = torch.empty_like(z)
    dgamma = torch.empty((N, c), dtype=torch.float32, device=z.device)
    dbeta = torch.empty((N, c), dtype=torch.float32, device=z.device)
    L.check(L.lib().vnqa_film_relu_res_bwd(L.ptr(dout), L.ptr(z), L.ptr(gamma), L.ptr(beta), L.ptr(dz), L.ptr(dgamma),
                                           L.ptr(dbeta), N, hp, wp, c, L.dtype_id(z.dtype), L.stream()),
            "vnqa_film_relu_res_bwd")
    return dz, dgamma, dbeta


def relu_bwd(a, y, b=None):
    """(a [+ b]) * [y > 0], all padded-NHWC tensors of the same shape/dtype."""
    g = torch.empty_like(a)
    L.check(L.lib().vnqa_relu_bwd(L.ptr(a), L.ptr(b), L.ptr(y), L.ptr(g), a.numel(), L.dtype_id(a.dtype), L.stream()),
            "vnqa_relu_bwd")
    return g


def conv3d_igemm(x, wt, bias=None, relu=False, pool2=False, out=None):
    """3-D conv (k=3, pad=1).  x: padded NDHWC [N, D+2, H+2, W+2, Cin]; wt [Cout][27][Cin];
    returns padded NDHWC [N, D+2, Ho+2, Wo+2, Cout] (pool2 pools (1,2,2))."""
    N, Dp, Hp, Wp, Cin = x.shape
    D, H, W = Dp - 2, Hp - 2, Wp - 2
    c_out, taps, cin_w = wt.shape
    assert taps == 27 and cin_w == Cin and wt.dtype == x.dtype
    Ho, Wo = (H // 2, W // 2) if pool2 else (H, W)
    if out is None:
        out = torch.zeros((N, Dp, Ho + 2, Wo + 2, c_out), dtype=x.dtype, device=x.device)
    # 128-cout 3-D convs (VideoOnlyCNN3D conv2 / conv3a): the 512 x 128 tile (tools/ab_c3d_tiles.sh: config 2 +1.5 % over the 256 x 128 default)
    tile = (15 if c_out == 128 else L.TILE_AUTO) if L.is_half(x.dtype) else L.TILE_AUTO      # (tools/ab_c3d_tiles.sh measured the ids)
    d = L.ConvDesc(L.dtype_id(x.dtype), N, H, W, Cin, c_out, out.shape[-1], 27, 1, 1, int(relu), 1 if pool2 else 0,
                   tile, 0, D)
    L.check(L.lib().vnqa_conv2d_igemm_fwd(ctypes.byref(d), L.ptr(x), L.ptr(wt), L.ptr(bias), None, None, L.ptr(out),
                                          L.stream()), "vnqa_conv2d_igemm_fwd(3d)")
    return out


def conv3d_wgrad(x, dy, want_bias=True):
    """x, dy: padded NDHWC of the same geometry.  Returns (dwt fp32 [Cout][27][Cin], dbias fp32 [Cout])."""
    N, Dp, Hp, Wp, Cin = x.shape
    Cout = dy.shape[-1]
    assert dy.shape[:4] == x.shape[:4] and dy.dtype == x.dtype
    d, h, w = Dp - 2, Hp - 2, Wp - 2
    ws = workspace(L.lib().vnqa_conv3d_wgrad_workspace(N, d, h, w, Cin, Cout), x.device)
    dwt = torch.empty((Cout, 27, Cin), dtype=torch.float32, device=x.device)
    dbias = torch.empty((Cout,), dtype=torch.float32, device=x.device) if want_bias else None
    L.check(L.lib().vnqa_conv3d_wgrad(L.ptr(x), L.ptr(dy), L.ptr(dwt), L.ptr(dbias), L.ptr(ws), N, d, h, w, Cin, Cout,
                                      L.dtype_id(x.dtype), L.stream()), "vnqa_conv3d_wgrad")
    return dwt, dbias


def temporal_attn_fwd(feat, valid, mask, w, bias):
    B, T, A = feat.shape
    coef = torch.empty((B, T), dtype=torch.float32, device=feat.device)
    ctxt = torch.empty((B, A), dtype=torch.float32, device=feat.device)
    L.check(L.lib().vnqa_temporal_attn_fwd(L.ptr(feat), L.ptr(valid), L.ptr(mask), L.ptr(w), L.ptr(bias), L.ptr(coef),
                                           L.ptr(ctxt), B, T, A, L.stream()), "vnqa_temporal_attn_fwd")
    return coef, ctxt


def temporal_attn_bwd(feat, valid, w, coef, dctxt):
    B, T, A = feat.shape
    dfeat = torch.empty_like(feat)
    dw_part = torch.empty((B, A), dtype=torch.float32, device=feat.device)
    db_part = torch.empty((B,), dtype=torch.float32, device=feat.device)
    L.check(L.lib().vnqa_temporal_attn_bwd(L.ptr(feat), L.ptr(valid), L.ptr(w), L.ptr(coef), L.ptr(dctxt), L.ptr(dfeat),
                                           L.ptr(dw_part), L.ptr(db_part), B, T, A, L.stream()), "vnqa_temporal_attn_bwd")
    return dfeat, dw_part, db_part


# ---- csrc/glue.hip: small fp32 products and data movement of the question path / classifier / loss ---------------------
def _f32c(t):
    assert t.dtype == torch.float32 and t.is_cuda
    return t


def sgemm(a, b, m, n, k, a_rs, a_cs, b_rs, b_cs, out, bias=None, relu=False, accumulate=False, a_mask=None, a_rows=None,
          c_rows=None):
    """out[row_c(i)][j] = act(sum_k a'[i,k] b[k,j] + bias[j]) [+ out]; element strides select NN / NT / TN (vnqa_sgemm)."""
    assert out.dtype == torch.float32 and out.stride(-1) == 1
    ws_bytes = L.lib().vnqa_sgemm_workspace(m, n, k)
    ws = workspace(ws_bytes, out.device) if ws_bytes > 0 else None
    L.check(L.lib().vnqa_sgemm(L.vptr(_f32c(a)), L.vptr(_f32c(b)), L.vptr(out), L.ptr(bias),
                               L.vptr(a_mask) if a_mask is not None else None, L.ptr(a_rows), L.ptr(c_rows),
                               a_rs, a_cs, b_rs, b_cs, out.stride(0), m, n, k, int(relu), int(accumulate), None, L.ptr(ws),
                               L.stream()), "vnqa_sgemm")
    return out


def linear_nt2(x, w, addend=None, out2_col=None, out2_mul=None):
    """(y, y2) with y = x @ w.T (+ addend) and, from the same epilogue, y2 = y * out2_col[n] * out2_mul[m, n] (vnqa_sgemm2)."""
    m, k = x.shape
    n = w.shape[0]
    y = torch.empty((m, n), dtype=torch.float32, device=x.device)
    y2 = torch.empty_like(y)
    ws_bytes = L.lib().vnqa_sgemm_workspace(m, n, k)
    ws = workspace(ws_bytes, x.device) if ws_bytes > 0 else None
    L.check(L.lib().vnqa_sgemm2(L.vptr(_f32c(x)), L.vptr(_f32c(w)), L.ptr(y), None, x.stride(0), 1, 1, w.stride(0), n, m, n, k,
                                L.ptr(addend), L.ptr(y2), L.ptr(out2_col), L.ptr(out2_mul), L.ptr(ws), L.stream()), "vnqa_sgemm2")
    return y, y2


def linear_nt(x, w, bias=None, relu=False, a_rows=None, m=None):
    """act(x_sel @ w.T + bias): x [R,K], w [N,K] fp32 (row-major, unit inner stride); a_rows gathers m rows of x."""
    m = x.shape[0] if m is None else m
    n, k = w.shape
    out = torch.empty((m, n), dtype=torch.float32, device=x.device)
    return sgemm(x, w, m, n, k, x.stride(0), 1, 1, w.stride(0), out, bias=bias, relu=relu, a_rows=a_rows)


def matmul_nn(a, b, a_mask=None, out=None, c_rows=None, accumulate=False):
    """(a * [mask > 0]) @ b: a [M,K], b [K,N]; c_rows scatters the result rows into `out`."""
    m, k = a.shape
    n = b.shape[1]
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    return sgemm(a, b, m, n, k, a.stride(0), 1, b.stride(0), 1, out, a_mask=a_mask, c_rows=c_rows, accumulate=accumulate)


def matmul_tn(a, b, a_mask=None, out=None, accumulate=False):
    """(a * [mask > 0]).T @ b: a [K,M], b [K,N] -> [M,N]."""
    k, m = a.shape
    n = b.shape[1]
    if out is None:
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    return sgemm(a, b, m, n, k, 1, a.stride(0), b.stride(0), 1, out, a_mask=a_mask, accumulate=accumulate)


def colsum(x, mask=None, out=None):
    """sum over the rows of a 2-D tensor (fp32, optionally only where mask > 0; or bf16 without a mask) -> fp32 [cols]."""
    rows, cols = x.shape
    if out is None:
        out = torch.empty((cols,), dtype=torch.float32, device=x.device)
    assert out.numel() == cols and out.is_contiguous() and out.dtype == torch.float32
    L.check(L.lib().vnqa_colsum(L.vptr(x), L.vptr(mask) if mask is not None else None, L.ptr(out), rows, cols,
                                x.stride(0), L.dtype_id(x.dtype), L.stream()), "vnqa_colsum")
    return out


def gather_rows(src, rows):
    """dst[r] = src[rows[r]] (zeros for negative indices); src [R,C] fp32 contiguous, rows int32."""
    n, c = rows.numel(), src.shape[1]
    dst = torch.empty((n, c), dtype=torch.float32, device=src.device)
    L.check(L.lib().vnqa_gather_rows(L.ptr(_f32c(src)), L.ptr(rows), L.ptr(dst), n, c, L.stream()), "vnqa_gather_rows")
    return dst


def embed_proj_fwd(tokens, row_perm, embed, w_ih, b_ih, b_hh):
    """Returns (xg [B,Lq,4H], rows int32 [B*Lq]: the token of every position, kept for the backward)."""
    B, Lq = tokens.shape
    V, E = embed.shape
    G = w_ih.shape[0]
    xg = torch.empty((B, Lq, G), dtype=torch.float32, device=embed.device)
    rows = torch.empty((B * Lq,), dtype=torch.int32, device=embed.device)
    L.check(L.lib().vnqa_embed_proj_fwd(L.ptr(tokens), L.ptr(row_perm), L.ptr(embed), L.ptr(w_ih), L.ptr(b_ih), L.ptr(b_hh),
                                        L.ptr(xg), L.ptr(rows), B, Lq, E, G, V, L.stream()), "vnqa_embed_proj_fwd")
    return xg, rows


def token_dsum(rows, dxg, vocab):
    B, Lq, G = dxg.shape
    dsum = torch.empty((vocab, G), dtype=torch.float32, device=dxg.device)
    L.check(L.lib().vnqa_token_dsum(L.ptr(rows), L.ptr(dxg), L.ptr(dsum), B * Lq, G, vocab, L.stream()), "vnqa_token_dsum")
    return dsum


def lstm_fold_dxg(dgates, q_lens_i32, Lq, n_rep):
    B, S, G = dgates.shape
    dxg = torch.empty((B, Lq, G), dtype=torch.float32, device=dgates.device)
    L.check(L.lib().vnqa_lstm_fold_dxg(L.ptr(dgates), L.ptr(q_lens_i32), L.ptr(dxg), B, Lq, S, G // 4, n_rep, L.stream()),
            "vnqa_lstm_fold_dxg")
    return dxg


def lstm_wgrad_operands(dgates, hs, h0, dtype):
    B, S, G = dgates.shape
    H = G // 4
    a = torch.empty((B * S, G), dtype=dtype, device=dgates.device)
    hp = torch.empty((B * S, H), dtype=dtype, device=dgates.device)
    L.check(L.lib().vnqa_lstm_wgrad_operands(L.ptr(dgates), L.ptr(hs), L.ptr(h0), L.ptr(a), L.ptr(hp), B, S, H,
                                             L.dtype_id(dtype), L.stream()), "vnqa_lstm_wgrad_operands")
    return a, hp


def ce_loss(logits, ys, row_perm, weight, mean):
    B, Kc = logits.shape
    assert ys.dtype == torch.int64 and ys.is_contiguous() and ys.numel() >= B, "ce_loss: targets must be contiguous int64"
    assert logits.dtype == torch.float32 and logits.is_contiguous()
    assert row_perm is None or (row_perm.dtype == torch.int32 and row_perm.is_contiguous())
    loss = torch.empty((), dtype=torch.float32, device=logits.device)
    dlogits = torch.empty_like(logits)
    L.check(L.lib().vnqa_ce_loss(L.ptr(logits), L.ptr(ys), L.ptr(row_perm), L.ptr(weight), L.ptr(loss), L.ptr(dlogits), B, Kc,
                                 int(mean), L.stream()), "vnqa_ce_loss")
    return loss, dlogits


def bn_running_update(mean, var, frame_off_i32, n_frames, pixels_per_image, running_mean, running_var, momentum):
    C = running_mean.numel()
    L.check(L.lib().vnqa_bn_running_update(L.ptr(mean), L.ptr(var), L.ptr(frame_off_i32), L.ptr(running_mean),
                                           L.ptr(running_var), n_frames, pixels_per_image, C, mean.stride(0), momentum,
                                           L.stream()), "vnqa_bn_running_update")


def temporal_attn_packed_fwd(f, frame_off_i32, n_frames, B, T, A, w, bias):
    coef = torch.empty((B, T), dtype=torch.float32, device=f.device)
    ctxt = torch.empty((B, A), dtype=torch.float32, device=f.device)
    L.check(L.lib().vnqa_temporal_attn_packed_fwd(L.ptr(f), f.stride(0), L.dtype_id(f.dtype), L.ptr(frame_off_i32), n_frames,
                                                  L.ptr(w), L.ptr(bias), L.ptr(coef), L.ptr(ctxt), B, T, A, L.stream()),
            "vnqa_temporal_attn_packed_fwd")
    return coef, ctxt


def temporal_attn_packed_bwd(f, frame_off_i32, n_frames, B, T, A, w, coef, dctxt, grad_scale=1.0):
    df = torch.empty_like(f)
    dw_part = torch.empty((B, A), dtype=torch.float32, device=f.device)
    db_part = torch.empty((B, 1), dtype=torch.float32, device=f.device)
    L.check(L.lib().vnqa_temporal_attn_packed_bwd(L.ptr(f), f.stride(0), L.dtype_id(f.dtype), L.ptr(frame_off_i32), n_frames,
                                                  L.ptr(w), L.ptr(coef), L.ptr(dctxt), L.ptr(df), L.ptr(dw_part),
                                                  L.ptr(db_part), B, T, A, float(grad_scale), L.stream()),
            "vnqa_temporal_attn_packed_bwd")
    return df, dw_part, db_part


def mac_wgrad(rows, d, tensors):
    """vnqa_mac_core_wgrad: every parameter gradient of the MAC reasoning steps from the step-stacked [rows, d] factors."""
    a = L.MacWgrad(rows, d)
    for name, t in tensors.items():
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous()), name
        setattr(a, name, None if t is None else t.data_ptr())
    L.check(L.lib().vnqa_mac_core_wgrad(ctypes.byref(a), L.stream()), "vnqa_mac_core_wgrad")


def mac_core_call(direction, dims, tensors, defer_wgrad=False):
    """vnqa_mac_core_fwd / _bwd: `dims` = (n, d, lq, s, ld, dtype id), `tensors` = {field name: tensor or None}."""
    a = L.MacCore(*dims)
    for name, t in tensors.items():
        setattr(a, name, None if t is None else t.data_ptr())
    a.defer_wgrad = 1 if defer_wgrad else 0
    fn = L.lib().vnqa_mac_core_fwd if direction == "fwd" else L.lib().vnqa_mac_core_bwd
    L.check(fn(ctypes.byref(a), L.stream()), "vnqa_mac_core_" + direction)


def mac_chain_call(direction, dims, n_steps, tensors, memories, mask_m, d_memory_out=None, d_concat=None):
    """vnqa_mac_chain_fwd / _bwd: `tensors` describe STEP 0 (views of the step-stacked slabs), see include/vnqa_hip.h."""
    a = L.MacCore(*dims)
    for name, t in tensors.items():
        setattr(a, name, None if t is None else t.data_ptr())
    a.defer_wgrad = 1
    if direction == "fwd":
        L.check(L.lib().vnqa_mac_chain_fwd(ctypes.byref(a), n_steps, L.ptr(memories), L.ptr(mask_m), L.stream()), "vnqa_mac_chain_fwd")
    else:
        L.check(L.lib().vnqa_mac_chain_bwd(ctypes.byref(a), n_steps, L.ptr(memories), L.ptr(mask_m), L.ptr(d_memory_out),
                                           L.ptr(d_concat), L.stream()), "vnqa_mac_chain_bwd")


# ---- csrc/cnn3d.hip: VideoOnlyCNN3D's first conv, BatchNorm over channel-last rows, MaxPool3d(4,4,4) ------------------------
def view_padded_ndhwc(D, H, W, C):
    """element (n, d, h, w, c) of a padded NDHWC tensor [N, D+2, H+2, W+2, C] (interior only)"""
    sh = (W + 2) * C
    sd = (H + 2) * sh
    return L.View5(sd + sh + C, (D + 2) * sd, sd, sh, C, 1, D, H, W)


def view_dense(D, H, W, C):
    return L.View5(0, D * H * W * C, H * W * C, W * C, C, 1, D, H, W)


def view_nc_flat(D, H, W, C):
    """[N, C*D*H*W] in NCDHW order (what .view(N, -1) of an NCDHW tensor gives: v_only_cnn3d.py:74)"""
    return L.View5(0, C * D * H * W, H * W, W, 1, D * H * W, D, H, W)


def bn_finalize(partial, count, eps, momentum=0.0, running_mean=None, running_var=None):
    """partial fp32 [blocks, C, 2] -> (mean, rstd) fp32 [C]; running statistics updated in place when given."""
    nblk, C, _ = partial.shape
    mean = torch.empty((C,), dtype=torch.float32, device=partial.device)
    rstd = torch.empty_like(mean)
    L.check(L.lib().vnqa_bn_finalize(L.ptr(partial), nblk, C, float(count), float(eps), float(momentum), L.ptr(mean), L.ptr(rstd),
                                     L.ptr(running_mean), L.ptr(running_var), L.stream()), "vnqa_bn_finalize")
    return mean, rstd


def c3d_stats_ncdhw(x):
    """fp32 [N, C, ...] -> partial statistics [blocks, C, 2]"""
    N, C = x.shape[0], x.shape[1]
    S = x[0, 0].numel()
    chunks = max(1, min(64, S // 16384))
    partial = torch.empty((N * chunks, C, 2), dtype=torch.float32, device=x.device)
    L.check(L.lib().vnqa_c3d_stats_ncdhw(L.ptr(_f32c(x)), L.ptr(partial), N, C, S, chunks, L.stream()), "vnqa_c3d_stats_ncdhw")
    return partial


def c3d_stats_rows(x):
    """dense [R, C] (fp32 or 16-bit) -> partial statistics [blocks, C, 2]"""
    R, C = x.shape
    assert x.is_contiguous()
    nb = L.lib().vnqa_c3d_stats_blocks(R)
    partial = torch.empty((nb, C, 2), dtype=torch.float32, device=x.device)
    L.check(L.lib().vnqa_c3d_stats_rows(L.vptr(x), L.ptr(partial), R, C, L.dtype_id(x.dtype), L.stream()), "vnqa_c3d_stats_rows")
    return partial


def bn_rows_apply(x, out, view, mean, rstd, gamma, beta):
    """out(view) = gamma (x - mean) rstd + beta; x dense [R, C]"""
    R, C = x.shape
    assert x.is_contiguous() and out.is_contiguous()
    L.check(L.lib().vnqa_bn_rows_apply(L.vptr(x), L.dtype_id(x.dtype), L.vptr(out), L.dtype_id(out.dtype), ctypes.byref(view),
                                       L.ptr(mean), L.ptr(rstd), L.ptr(_f32c(gamma)), L.ptr(_f32c(beta)), R, C, L.stream()),
            "vnqa_bn_rows_apply")
    return out


def bn_rows_bwd(dy, dy_view, x, dx_dtype, mean, rstd, gamma, grad_scale=1.0):
    """train-mode BatchNorm backward over rows: returns (dx dense [R, C] in dx_dtype, dgamma, dbeta)."""
    R, C = x.shape
    assert x.is_contiguous() and dy.is_contiguous()
    nb = L.lib().vnqa_c3d_stats_blocks(R)
    ws = workspace((nb * C * 2 + 2 * C) * 4, x.device)
    dx = torch.empty((R, C), dtype=dx_dtype, device=x.device)
    dg = torch.empty((C,), dtype=torch.float32, device=x.device)
    db = torch.empty_like(dg)
    L.check(L.lib().vnqa_bn_rows_bwd(L.vptr(dy), L.dtype_id(dy.dtype), ctypes.byref(dy_view), L.vptr(x), L.dtype_id(x.dtype),
                                     L.vptr(dx), L.dtype_id(dx_dtype), L.ptr(mean), L.ptr(rstd), L.ptr(_f32c(gamma)), L.ptr(dg),
                                     L.ptr(db), L.ptr(ws), float(grad_scale), R, C, L.stream()), "vnqa_bn_rows_bwd")
    return dx, dg, db


def pool444_fwd(y):
    """y padded NDHWC [N, D+2, H+2, W+2, C] (16-bit, ReLU applied) -> (p dense [N, D/4, H/4, W/4, C], idx uint8, partial stats)"""
    N, Dp, Hp, Wp, C = y.shape
    D, H, W = Dp - 2, Hp - 2, Wp - 2
    p = torch.empty((N, D // 4, H // 4, W // 4, C), dtype=y.dtype, device=y.device)
    idx = torch.empty(p.shape, dtype=torch.uint8, device=y.device)
    nb = L.lib().vnqa_pool444_blocks(N, D, H, W, C)
    partial = torch.empty((nb, C, 2), dtype=torch.float32, device=y.device)
    L.check(L.lib().vnqa_pool444_fwd(L.ptr(y), L.ptr(p), L.vptr(idx), L.ptr(partial), N, D, H, W, C, L.stream()), "vnqa_pool444_fwd")
    return p, idx, partial


def pool444_bwd(dp, idx, dy):
    """dy (padded NDHWC, zero halo kept) <- dp routed to the arg-max positions, zeros elsewhere in the interior"""
    N, Dp, Hp, Wp, C = dy.shape
    assert dp.is_contiguous() and idx.is_contiguous() and dp.dtype == dy.dtype
    L.check(L.lib().vnqa_pool444_bwd(L.ptr(dp), L.vptr(idx), L.ptr(dy), N, Dp - 2, Hp - 2, Wp - 2, C, L.stream()), "vnqa_pool444_bwd")
    return dy


def c3d_conv1_supported(N, D, H, W):
    return bool(L.lib().vnqa_c3d_conv1_supported(N, D, H, W))


def c3d_conv1_fwd(x, weight, bias, mean, rstd, gamma, beta, dtype):
    """fp32 clip [N,3,D,H,W] -> (p dense [N,D,H/2,W/2,64] in `dtype`, idx uint8, partial stats of p)"""
    N, _, D, H, W = x.shape
    p = torch.empty((N, D, H // 2, W // 2, 64), dtype=dtype, device=x.device)
    idx = torch.empty(p.shape, dtype=torch.uint8, device=x.device)
    partial = torch.empty((L.lib().vnqa_c3d_conv1_fwd_blocks(N, H, W), 64, 2), dtype=torch.float32, device=x.device)
    L.dtype_id(dtype)
    L.check(L.lib().vnqa_c3d_conv1_fwd(L.ptr(_f32c(x)), L.ptr(_f32c(weight)), L.ptr(_f32c(bias)), L.ptr(mean), L.ptr(rstd),
                                       L.ptr(_f32c(gamma)), L.ptr(_f32c(beta)), L.ptr(p), L.vptr(idx), L.ptr(partial), N, D, H, W,
                                       L.stream()), "vnqa_c3d_conv1_fwd")
    return p, idx, partial


def c3d_conv1_bwd(x, weight, mean, rstd, gamma, beta, dp, idx, grad_scale=1.0):
    """-> (dweight [64,3,3,3,3], dbias [64], bn_input dgamma [3], dbeta [3])"""
    N, _, D, H, W = x.shape
    ws = workspace((2 * L.lib().vnqa_c3d_conv1_bwd_blocks(N, H, W) + 16) * 64 * 112 * 4, x.device)
    dw = torch.empty((64, 3, 3, 3, 3), dtype=torch.float32, device=x.device)
    db = torch.empty((64,), dtype=torch.float32, device=x.device)
    dg = torch.empty((3,), dtype=torch.float32, device=x.device)
    dbt = torch.empty((3,), dtype=torch.float32, device=x.device)
    assert dp.is_contiguous() and idx.is_contiguous()
    L.check(L.lib().vnqa_c3d_conv1_bwd(L.ptr(_f32c(x)), L.ptr(_f32c(weight)), L.ptr(mean), L.ptr(rstd), L.ptr(_f32c(gamma)),
                                       L.ptr(_f32c(beta)), L.ptr(dp), L.vptr(idx), L.ptr(ws), float(grad_scale), L.ptr(dw), L.ptr(db),
                                       L.ptr(dg), L.ptr(dbt), N, D, H, W, L.stream()), "vnqa_c3d_conv1_bwd")
    return dw, db, dg, dbt
