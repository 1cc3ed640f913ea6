"""Calibrated rounding of the FROZEN stem's 16-bit weights.

A frozen layer's fp32 weights must become 16-bit values once; round-to-nearest makes each weight's error independent, so an output
channel's error  e(p) = sum_j x_j(p) (w16_j - w_j)  is a random walk over its K = c_in * taps terms — and the part of it that is
COHERENT over pixels (the input patches x(p) have a large common component: post-ReLU means, neighbouring taps that see the same
values) is a per-channel offset that no later pooling or averaging removes.  Here the rounding direction of every weight is chosen
so that E_p[e(p)^2] = d^T H d is small, with H = E[x x^T] the second moment of the layer's input patches over calibration frames —
the GPTQ recipe (Frantar et al. 2022: round one column, spread its error over the not-yet-rounded columns through H^-1), with the
grid being the 16-bit FLOAT grid and every weight confined to the two 16-bit neighbours of its fp32 value (so the element-wise
bound of round-to-nearest, one ulp instead of half, survives on any input).  Nothing changes at run time: same kernels, same bytes.

The statistics come from an fp32 torch pass of calibration frames through the reference layer sequence (models/obj_detector.py:69-86
behind VGG-16 features[0:10]); bench.py / the tests use seeded uniform-noise frames (the benchmark's data is synthetic noise);
a deployment passes frames of its own videos."""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _neighbours(w, dtype):
    """(nearest, other) 16-bit neighbours of fp32 w as fp32 tensors; other == nearest where w is exactly representable."""
    r = w.to(dtype)
    rf = r.float()
    bits = r.view(torch.int16).to(torch.int32)
    up = rf < w
    mag_up = (rf > 0) | ((rf == 0) & up)
    ob = bits + torch.where(up == mag_up, torch.ones_like(bits), -torch.ones_like(bits))
    ob = torch.where(rf == 0, torch.where(up, torch.ones_like(bits), torch.full_like(bits, -32767)), ob)
    other = ob.to(torch.int16).view(dtype).float()
    other = torch.where((rf == w) | ~torch.isfinite(other), rf, other)
    return rf, other


@torch.no_grad()
def stem_layer_inputs(vgg, od, frames):
    """fp32 torch pass of `frames` [N, 3, H, W] through the stem; returns {layer key: that layer's INPUT activation [N, C, h, w]}
    (keys: first = conv1_1, vgg0 = conv1_2, vgg1 = conv2_1, vgg2 = conv2_2, od0 = conv11 and the composed 5x5, od1 = conv12,
    od2 = conv21, od3 = conv22, od4 = conv31, od5 = conv32)."""
    f = vgg.features
    conv = lambda t, c: F.conv2d(t, c.weight.float(), c.bias.float(), padding=1)
    bn = lambda t, b: F.batch_norm(t, b.running_mean.float(), b.running_var.float(), b.weight.float(), b.bias.float(), False, 0.0, BN_EPS)
    x = frames.float().to(f["0"].weight.device)
    a = {"first": x}
    a["vgg0"] = F.relu(conv(a["first"], f["0"]))
    a["vgg1"] = F.max_pool2d(F.relu(conv(a["vgg0"], f["2"])), 2)
    a["vgg2"] = F.relu(conv(a["vgg1"], f["5"]))
    a["od0"] = bn(F.max_pool2d(F.relu(conv(a["vgg2"], f["7"])), 2), od.bn_input)
    a["od1"] = conv(a["od0"], od.conv11)
    a["od2"] = F.max_pool2d(F.relu(bn(conv(a["od1"], od.conv12), od.bn1)), 2)
    a["od3"] = conv(a["od2"], od.conv21)
    a["od4"] = F.max_pool2d(F.relu(bn(conv(a["od3"], od.conv22), od.bn2)), 2)
    a["od5"] = conv(a["od4"], od.conv31)
    return a


@torch.no_grad()
def second_moment(a, ksize, max_patches=60000, seed=0):
    """H = E[x x^T] [K, K] (float64) over the ksize x ksize 'same'-padded patches x (K = C * ksize^2, channel-major like
    weight.reshape(c_out, -1)) of activation a [N, C, h, w]; at most max_patches patches, drawn evenly from the frames."""
    N, C, h, w = a.shape
    K = C * ksize * ksize
    per = max(1, min(h * w, max_patches // N))
    g = torch.Generator().manual_seed(seed)
    H = torch.zeros(K, K, dtype=torch.float64, device=a.device)
    n = 0
    for i in range(N):
        cols = F.unfold(a[i:i + 1], ksize, padding=ksize // 2)[0]            # [K, h*w]
        if per < h * w:
            cols = cols[:, torch.randperm(h * w, generator=g)[:per].to(a.device)]
        cols = cols.double()
        H += cols @ cols.t()
        n += cols.shape[1]
    return H / n


@torch.no_grad()
def calibrated_round(w, H, dtype, damp=0.01):
    """w [c_out, c_in, kh, kw] fp32 -> fp32 values exactly representable in `dtype`, each one of the two 16-bit neighbours of its
    original, chosen column by column (largest H diagonal first) with the rounding error of every column fed forward through H^-1."""
    co = w.shape[0]
    W = w.detach().double().reshape(co, -1).clone()
    K = W.shape[1]
    H = H.to(W.device).double().clone()
    near, other = _neighbours(w.detach().float().reshape(co, -1), dtype)
    lo, hi = torch.minimum(near, other).double(), torch.maximum(near, other).double()
    d = torch.diagonal(H)
    dead = d <= 0
    H[dead, dead] = 1.0
    W[:, dead] = w.detach().double().reshape(co, -1)[:, dead]
    perm = torch.argsort(torch.diagonal(H), descending=True)
    W, lo, hi, H = W[:, perm], lo[:, perm], hi[:, perm], H[perm][:, perm]
    H += damp * torch.diagonal(H).mean() * torch.eye(K, dtype=H.dtype, device=H.device)
    L_ = torch.linalg.cholesky(H)
    U = torch.linalg.cholesky(torch.cholesky_inverse(L_), upper=True)        # H^-1 = U^T U
    Q = torch.empty_like(W)
    B = 128
    for j0 in range(0, K, B):
        j1 = min(j0 + B, K)
        Wb = W[:, j0:j1].clone()
        Eb = torch.empty_like(Wb)
        Ub = U[j0:j1, j0:j1]
        for j in range(j1 - j0):
            wj = Wb[:, j]
            q = torch.minimum(torch.maximum(wj.float().to(dtype).double(), lo[:, j0 + j]), hi[:, j0 + j])
            Q[:, j0 + j] = q
            e = (wj - q) / Ub[j, j]
            Wb[:, j:] -= e.unsqueeze(1) * Ub[j, j:].unsqueeze(0)
            Eb[:, j] = e
        W[:, j1:] -= Eb @ U[j0:j1, j1:]
    out = torch.empty_like(Q)
    out[:, perm] = Q
    return out.float().view_as(w)


@torch.no_grad()
def stem_calibration(vgg, od, frames=None, n_frames=48, height=224, width=224, seed=4242):
    """{layer key: H} for every stem layer (and 'od0c' = the composed 5x5 pair's 25-tap patches) from calibration frames
    [N, 3, H, W] in [0, 1] (default: seeded uniform noise, the benchmark's kind of data)."""
    if frames is None:
        frames = torch.rand(n_frames, 3, height, width, generator=torch.Generator().manual_seed(seed))
    acts = stem_layer_inputs(vgg, od, frames)
    out = {}
    for k, a in acts.items():
        # (the full-resolution layers have 50 176 patches per frame: a few frames of them suffice)
        a = a[:max(1, min(a.shape[0], 240000 // (a.shape[2] * a.shape[3]) + 1))]
        out[k] = second_moment(a, 3)
        if k == "od0":
            out["od0c"] = second_moment(a, 5)
    return out
