"""Random parameters of the MakeCutouts augmentations (main.py:164-198) and their composition for the HIP kernels.

kornia 0.5.10 is not available offline: the samplers below restate kornia/augmentation/random_generator (distributions, the
`_range_bound` treatment of scalar arguments) and the coordinate conventions of `warp_affine` / `warp_perspective`; an independent
CPU restatement of the operators themselves lives in oracle/kornia_aug.py (test infrastructure) and
tools/augment_deviation.py measures this module's fused form against it (profiles/r04_augment_deviation.txt).  Parity unpinned.

Two stages:

  draw_chain(N, S, augs, ...)   raw per-cutout draws, one dict per operator, in list order (the same dict layout the oracle's
                                `apply_chain` consumes, so both can be driven by the same draws)
  plan(chain, ...)              -> segments.  A *fused* segment is ONE launch of ffvc_augment_fwd:
                                    out(x) = erase( jitter( C * m(p1) * bilinear(src, clamp(A^-1 p1)) + c0 ) ),  p1 = P^-1 x
                                with A the border-padded affine slot (kornia RandomAffine(padding_mode='border') as the first
                                geometric operator), P the composed homography of every further geometric operator (zero padding
                                with grid_sample's one-pixel linear fade m when P rotates / shears / projects, plain clamping when
                                it only resizes or crops), `jitter` kornia's ColorJitter (hsv round trips,
                                clamps, random order: csrc/augment_cj.h), then the erase rectangle.  'Sh' / 'Et' / 'Ts' are their
                                own image -> image kernels (csrc/augment_ops.hip) between fused segments.  A new segment starts
                                wherever the list order leaves geometry -> colour -> erase (e.g. a warp after a jitter), so the
                                order of the reference's nn.Sequential is kept; `sequential=True` (the default: what the
                                reference computes) additionally gives every resampling operator its own pass (kornia's
                                sequential bilinear resamples: two interpolations for Af -> Pe); `sequential=False` composes
                                consecutive warps into one interpolation (one launch for the default set; opt-in, measurably not
                                kornia's result: profiles/r04_augment_deviation.txt).
  draw_params(...)              the single-launch (fused, sequential=False) form: plan(draw_chain(...)) flattened — the kernel's
                                own formula, used by the kernel-level tests.

Operators (probability p per sample unless noted; kornia names in main.py:164-198):
    'Af'  RandomAffine(degrees=15, translate=0.1, p=0.7, padding_mode='border')   angle U(-15,15) deg about ((W-1)/2,(H-1)/2);
          translate=0.1 is a scalar: `_range_bound(0.1, bounds=(0,1))` = (0, 0.1) read as (max_dx, max_dy) fractions ->
          dx = 0, dy ~ U(-0.1 H, 0.1 H)
    'Pe'  RandomPerspective(distortion_scale=0.7, p=0.7)      every corner moves inwards by U(0, 0.35*size) per axis
    'Ro'  RandomRotation(degrees=15, p=0.7)                   angle U(-15,15) deg about the centre, zero padding
    'Re'  RandomResizedCrop(scale=(0.1,1), ratio=(3/4,4/3), p=1)   crop of area U(.1,1)*S^2, log-uniform aspect, resized
    'Re2' RandomResizedCrop(scale=(0.9,1), ...)
    'R'   Resize(cut_size): bilinear, align_corners=False (main.py:145-152) — identity when the source already has cut_size
    'Cr'  RandomCrop(cut_size, p=0.5), 'Cc' CenterCrop(cut_size)   identities when the source already has cut_size; on a
          larger source (pool_size > cut_size or pool=False) a random / centred integer window (always taken: a batch cannot mix sizes)
    'Ji'  ColorJitter(hue=0.1, saturation=0.1, p=0.7)         hue U(-.1,.1) turns, saturation U(.9,1.1), random order
    'Ji2' ColorJitter(brightness=.1, contrast=.1, saturation=.05, hue=.05, p=0.5)
    'Er'  RandomErasing((.1,.4), (.3,1/.3), same_on_batch=True, p=0.7)    ONE rectangle (and one coin flip) per batch; area
          U(.1,.4)*S^2, aspect a 50/50 mixture of U(.3,1) and U(1,1/.3)
    'Er2' the same with same_on_batch=False
    'Gn'  RandomGaussianNoise(std=1, p=0.5)                   per-sample N(0,1) noise, merged with MakeCutouts' own
                                                              U(0,noise_fac)*N(0,1) term (sum of Gaussians)
    'Sh'  RandomSharpness(sharpness=0.4, p=0.7)               factor U(0.6, 1.4)
    'Et'  RandomElasticTransform(p=0.7)                       63x63 Gaussian (sigma 32) of U(-1,1) noise, alpha 1, normalised units
    'Ts'  RandomThinPlateSpline(scale=0.3, p=0.7)             5 control points moved by U(-0.3, 0.3)
The chain tracks the current image side: it starts at `src_size` (pool_size, or the raw image side with pool=False) and becomes
cut_size after 'R' / 'Re' / 'Re2' / 'Cr' / 'Cc' (out_size()).
"""
import math

import torch

SUPPORTED = ("Af", "Pe", "Ji", "Er", "Ro", "Re", "Re2", "Cr", "Cc", "Ji2", "Er2", "Gn", "R", "Sh", "Et", "Ts")
RESIZING = ("R", "Re", "Re2", "Cr", "Cc")
GEOMETRIC = ("Af", "Pe", "Ro", "Re", "Re2", "R", "Cr", "Cc")
DENSE = ("Sh", "Et", "Ts")
DEFAULT = ("Af", "Pe", "Ji", "Er")
_F64 = torch.float64


def _homography(src, dst):
    """Batched DLT: H (N,3,3) with dst ~ H src for 4 point pairs. src, dst: (N,4,2) float64."""
    N = src.shape[0]
    x, y, u, v = src[..., 0], src[..., 1], dst[..., 0], dst[..., 1]
    zeros, ones = torch.zeros_like(x), torch.ones_like(x)
    a1 = torch.stack([x, y, ones, zeros, zeros, zeros, -u * x, -u * y], dim=-1)
    a2 = torch.stack([zeros, zeros, zeros, x, y, ones, -v * x, -v * y], dim=-1)
    A = torch.cat([a1, a2], dim=1)                     # (N,8,8)
    b = torch.cat([u, v], dim=1).unsqueeze(-1)         # (N,8,1)
    h = torch.linalg.solve(A, b).squeeze(-1)
    return torch.cat([h, torch.ones(N, 1, dtype=h.dtype)], dim=1).view(N, 3, 3)


def _rot_fwd(angle_deg, translations, center):
    """(N,3,3) forward (source -> destination pixel) matrix of kornia's get_affine_matrix2d without scale / shear:
    rotation by `angle` about `center`, then the translation."""
    th = torch.deg2rad(angle_deg)
    cs, sn = torch.cos(th), torch.sin(th)
    cx, cy = center[:, 0], center[:, 1]
    M = torch.zeros(angle_deg.shape[0], 3, 3, dtype=_F64)
    M[:, 0, 0], M[:, 0, 1], M[:, 0, 2] = cs, -sn, cx - cs * cx + sn * cy + translations[:, 0]
    M[:, 1, 0], M[:, 1, 1], M[:, 1, 2] = sn, cs, cy - sn * cx - cs * cy + translations[:, 1]
    M[:, 2, 2] = 1.0
    return M


def out_size(S, augs, src_size=None):
    """Side of the augmented batch: cut_size once the chain holds a resize / crop, else the source's side."""
    return S if (src_size is None or src_size == S or any(a in RESIZING for a in augs)) else src_size


def draw_chain(N, S, augs=DEFAULT, generator=None, p=0.7, src_size=None, device=None):
    """Raw draws of every operator of the chain, in list order: [(name, dict)].  `on` (N,) bool = the operator's Bernoulli draw.
    S = cut_size, src_size = side of the image the chain starts from.  'Et' draws its (N,2,side,side) noise on `device`
    (default CPU) with torch's global generator of that device when `generator` lives elsewhere."""
    for a in augs:
        if a not in SUPPORTED:
            raise NotImplementedError(f"augmentation '{a}' is not built on the HIP path (built: {SUPPORTED})")
    g = generator
    rnd = lambda *s: torch.rand(*s, generator=g, dtype=_F64)  # noqa: E731
    uni = lambda n, lo, hi: lo + (hi - lo) * rnd(n)           # noqa: E731
    cut = S
    side = int(src_size or cut)                                # current side of the image as the chain advances
    chain = []
    for a in augs:
        if a in ("Af", "Ro"):
            d = dict(on=rnd(N) < p, angle=uni(N, -15.0, 15.0))
            if a == "Af":         # translate=0.1 -> (max_dx, max_dy) = (0, 0.1) * side  (see the module docstring)
                d["translations"] = torch.stack([torch.zeros(N, dtype=_F64), uni(N, -0.1 * side, 0.1 * side)], dim=1)
            else:
                d["translations"] = torch.zeros(N, 2, dtype=_F64)
            d["center"] = torch.full((N, 2), (side - 1) / 2.0, dtype=_F64)
            d["side"] = side
            chain.append((a, d))
        elif a == "Pe":
            start = torch.tensor([[0.0, 0.0], [side - 1.0, 0.0], [side - 1.0, side - 1.0], [0.0, side - 1.0]], dtype=_F64)
            sign = torch.tensor([[1.0, 1.0], [-1.0, 1.0], [-1.0, -1.0], [1.0, -1.0]], dtype=_F64)
            rv = rnd(N, 4, 2)
            chain.append((a, dict(on=rnd(N) < p, start=start[None].repeat(N, 1, 1), end=start[None] + 0.7 * side / 2.0 * rv * sign[None],
                                  side=side)))
        elif a in ("Re", "Re2"):
            lo = 0.1 if a == "Re" else 0.9
            area = uni(N, lo, 1.0) * side * side
            ratio = torch.exp(uni(N, math.log(0.75), math.log(4 / 3)))
            w = torch.sqrt(area * ratio).clamp(1, side)
            h = torch.sqrt(area / ratio).clamp(1, side)
            chain.append((a, dict(on=torch.ones(N, dtype=torch.bool), x0=rnd(N) * (side - w), y0=rnd(N) * (side - h), w=w, h=h, side=side,
                                  cut=cut)))
            side = cut
        elif a == "R":
            if side != cut:
                chain.append((a, dict(on=torch.ones(N, dtype=torch.bool), side=side, cut=cut)))
                side = cut
        elif a in ("Cr", "Cc"):
            if side < cut:
                raise ValueError(f"'{a}': the image ({side}) is smaller than cut_size ({cut})")
            if side > cut:                                     # integer window of the current image
                if a == "Cr":
                    x0, y0 = (rnd(N) * (side - cut + 1)).floor(), (rnd(N) * (side - cut + 1)).floor()
                else:
                    x0 = y0 = torch.full((N,), float((side - cut) // 2), dtype=_F64)
                chain.append((a, dict(on=torch.ones(N, dtype=torch.bool), x0=x0, y0=y0, side=side, cut=cut)))
                side = cut
        elif a in ("Ji", "Ji2"):
            br, ct, sa, hu, pj = (0.0, 0.0, 0.1, 0.1, p) if a == "Ji" else (0.1, 0.1, 0.05, 0.05, 0.5)
            chain.append((a, dict(on=rnd(N) < pj, brightness=uni(N, 1 - br, 1 + br), contrast=uni(N, 1 - ct, 1 + ct),
                                  saturation=uni(N, 1 - sa, 1 + sa), hue=uni(N, -hu, hu), order=torch.randperm(4, generator=g))))
        elif a in ("Er", "Er2"):
            same = a == "Er"
            n = 1 if same else N
            area = uni(n, 0.1, 0.4) * side * side
            r1, r2 = uni(n, 0.3, 1.0), uni(n, 1.0, 1 / 0.3)
            ratio = torch.where(rnd(n).round().bool(), r1, r2)
            hh = torch.sqrt(area * ratio).round().clamp(1, side)
            ww = torch.sqrt(area / ratio).round().clamp(1, side)
            xs, ys = (rnd(n) * (side - ww + 1)).floor(), (rnd(n) * (side - hh + 1)).floor()
            on = rnd(n) < p
            ex = (lambda t: t.expand(N).clone()) if same else (lambda t: t)
            chain.append((a, dict(on=ex(on), xs=ex(xs).long(), ys=ex(ys).long(), widths=ex(ww).long(), heights=ex(hh).long())))
        elif a == "Gn":
            chain.append((a, dict(on=rnd(N) < 0.5)))
        elif a == "Sh":
            chain.append((a, dict(on=rnd(N) < p, factor=uni(N, 0.6, 1.4))))
        elif a == "Et":
            dev = torch.device(device or "cpu")
            if g is not None and g.device == dev:
                noise = torch.rand(N, 2, side, side, generator=g, device=dev) * 2 - 1
            else:
                noise = torch.rand(N, 2, side, side, device=dev) * 2 - 1
            chain.append((a, dict(on=rnd(N) < p, noise=noise)))
        elif a == "Ts":
            src = torch.tensor([[-1.0, -1.0], [-1.0, 1.0], [1.0, -1.0], [1.0, 1.0], [0.0, 0.0]], dtype=_F64)[None].repeat(N, 1, 1)
            chain.append((a, dict(on=rnd(N) < p, src=src, dst=src + uni(N * 10, -0.3, 0.3).view(N, 5, 2))))
    return chain


def _conj_affine(M, side):
    """Pixel map of kornia's warp_affine(M, align_corners=False): normalize_homography uses the align_corners=True pixel
    convention, affine_grid / grid_sample the align_corners=False one, so the output pixel i is taken from
        x_src = T2 . M^-1 . T1 (i),   T1(i) = (i + 0.5) (W-1)/W,   T2(a) = a W/(W-1) - 0.5
    -> returns the FORWARD equivalent (source -> destination) T1^-1 . M . T2^-1."""
    s = side / max(side - 1.0, 1e-9)
    T1 = torch.tensor([[1 / s, 0, 0.5 / s], [0, 1 / s, 0.5 / s], [0, 0, 1]], dtype=_F64)
    T2 = torch.tensor([[s, 0, -0.5], [0, s, -0.5], [0, 0, 1]], dtype=_F64)
    return torch.linalg.inv(T1)[None] @ M @ torch.linalg.inv(T2)[None]


def _conj_perspective(M, side):
    """warp_perspective(M, align_corners=False): the grid is create_meshgrid(normalized) (align_corners=True style), the sampling
    align_corners=False: x_src = T2 . M^-1 (i)  ->  forward equivalent M . T2^-1."""
    s = side / max(side - 1.0, 1e-9)
    T2 = torch.tensor([[s, 0, -0.5], [0, s, -0.5], [0, 0, 1]], dtype=_F64)
    return M @ torch.linalg.inv(T2)[None]


def _fwd_matrix(name, d, N):
    """Forward pixel homography (source -> destination) of one geometric operator, identity where `on` is False."""
    eye = torch.eye(3, dtype=_F64).repeat(N, 1, 1)
    if name == "Af":
        M = _conj_affine(_rot_fwd(d["angle"], d["translations"], d["center"]), d["side"])
    elif name == "Ro":            # RandomRotation samples with align_corners=True: the pixel matrix as it stands
        M = _rot_fwd(d["angle"], d["translations"], d["center"])
    elif name == "Pe":
        M = _conj_perspective(_homography(d["start"], d["end"]), d["side"])
    elif name in ("Re", "Re2"):
        cut = d["cut"]
        M = torch.zeros(N, 3, 3, dtype=_F64)                 # crop [x0, x0+w-1] x [y0, y0+h-1] -> [0, cut-1]^2
        M[:, 0, 0] = (cut - 1) / (d["w"] - 1).clamp_min(1e-6)
        M[:, 1, 1] = (cut - 1) / (d["h"] - 1).clamp_min(1e-6)
        M[:, 0, 2] = -d["x0"] * M[:, 0, 0]
        M[:, 1, 2] = -d["y0"] * M[:, 1, 1]
        M[:, 2, 2] = 1.0
    elif name == "R":                                        # x_out = (x_in + .5) * cut / S - .5  (align_corners=False)
        M = torch.zeros(N, 3, 3, dtype=_F64)
        M[:, 0, 0] = M[:, 1, 1] = d["cut"] / d["side"]
        M[:, 0, 2] = M[:, 1, 2] = 0.5 * d["cut"] / d["side"] - 0.5
        M[:, 2, 2] = 1.0
    elif name in ("Cr", "Cc"):
        M = eye.clone()
        M[:, 0, 2], M[:, 1, 2] = -d["x0"], -d["y0"]
    else:
        raise ValueError(name)
    return torch.where(d["on"][:, None, None], M, eye)


def tps_params(src, dst):
    """(N,5,2) control points -> (N,26) fp32 rows for ffvc_tps_grid: kornia's get_tps_transform(dst, src) (the spline takes the
    destination grid to source locations): centres = dst... see oracle/kornia_aug.py::thin_plate_spline."""
    N, P, _ = src.shape
    s, d = dst.to(_F64), src.to(_F64)                         # get_tps_transform(points_src=dst, points_dst=src)
    d2 = (s[:, :, None, :] - d[:, None, :, :]).pow(2).sum(-1)  # kernel centred on points_dst (= src), evaluated at points_src
    k = 0.5 * d2 * torch.log(d2 + 1e-8)
    ones = torch.ones(N, P, 1, dtype=_F64)
    pm = torch.cat([ones, s], dim=-1)
    L = torch.cat([torch.cat([k, pm], dim=-1), torch.cat([pm.transpose(1, 2), torch.zeros(N, 3, 3, dtype=_F64)], dim=-1)], dim=1)
    w = torch.linalg.solve(L, torch.cat([d, torch.zeros(N, 3, 2, dtype=_F64)], dim=1))
    centres = src.to(_F64)                                    # warp_image_tps(image, kernel_centers=src, ...)
    return torch.cat([centres.reshape(N, 10), w[:, :5].reshape(N, 10), w[:, 5].reshape(N, 2), w[:, 6].reshape(N, 2),
                      w[:, 7].reshape(N, 2)], dim=1).float().contiguous()


def plan(chain, N, S, src_size=None, sequential=True):
    """-> list of segments: ("fused", params dict with pinv/ainv/cmat/coff/cj/erase/gn + "src"/"out" sides) or
    (name in DENSE, kernel parameters: on (N,) fp32 + factor | noise | tps).  See the module docstring."""
    side = int(src_size or S)
    segs = []
    cur = None

    def new_seg(src_side):
        return dict(src=src_side, out=src_side, stage=0, A=None, H=torch.eye(3, dtype=_F64).repeat(N, 1, 1), cj=None,
                    erase=torch.zeros(N, 4, dtype=torch.int32), gn=torch.zeros(N, dtype=_F64), n_geo=0, has_erase=False)

    def close():
        nonlocal cur
        if cur is not None:
            segs.append(("fused", _finish(cur, N)))
            cur = None

    for name, d in chain:
        if name in DENSE:
            close()
            if not segs:                                   # a dense operator first: the cutn-fold repeat is an identity launch
                segs.append(("fused", _finish(new_seg(side), N)))
            on = d["on"].float()
            if name == "Sh":
                segs.append((name, {"on": on, "factor": d["factor"].float()}))
            elif name == "Et":
                segs.append((name, {"on": on, "noise": d["noise"].float()}))
            else:
                segs.append((name, {"on": on, "tps": tps_params(d["src"], d["dst"])}))
            continue
        stage = 0 if name in GEOMETRIC else (1 if name in ("Ji", "Ji2") else (2 if name in ("Er", "Er2") else 3))
        need_new = cur is None or stage < cur["stage"] or (stage == 1 and cur["cj"] is not None) or (stage == 2 and cur["has_erase"])
        if name in GEOMETRIC and cur is not None and not need_new:
            # a border-padded affine only keeps its padding as the FIRST warp of a segment; `sequential`: one warp per pass
            # (kornia's RandomResizedCrop and Resize interpolate too; the integer crops 'Cr' / 'Cc' do not: composing them is exact)
            if (name == "Af" and cur["n_geo"] > 0) or (sequential and cur["n_geo"] > 0 and name in ("Af", "Pe", "Ro", "Re", "Re2", "R")):
                need_new = True
        if need_new:
            close()
            cur = new_seg(side)
        cur["stage"] = max(cur["stage"], stage)
        if name in GEOMETRIC:
            M = _fwd_matrix(name, d, N)
            if name == "Af" and cur["n_geo"] == 0:
                cur["A"] = M
            else:
                cur["H"] = M @ cur["H"]
            cur["n_geo"] += 1
            if name in RESIZING:
                side = d["cut"]
                cur["out"] = side
        elif name in ("Ji", "Ji2"):
            order = [int(k) for k in d["order"]]
            code = float(order[0] + 4 * order[1] + 16 * order[2] + 64 * order[3])
            cj = torch.zeros(N, 8, dtype=_F64)
            cj[:, 0] = d["on"].to(_F64)
            cj[:, 1], cj[:, 2], cj[:, 3], cj[:, 4], cj[:, 5] = d["brightness"], d["contrast"], d["saturation"], d["hue"], code
            cur["cj"] = cj
        elif name in ("Er", "Er2"):
            rect = torch.stack([d["xs"], d["ys"], d["xs"] + d["widths"], d["ys"] + d["heights"]], dim=1).to(torch.int32)
            cur["erase"] = torch.where(d["on"][:, None], rect, cur["erase"])
            cur["has_erase"] = True
        elif name == "Gn":
            cur["gn"] = torch.where(d["on"], torch.ones(N, dtype=_F64), cur["gn"])
    close()
    if not segs or segs[-1][0] != "fused":               # the last launch writes the patch rows (+ noise): an identity resample
        segs.append(("fused", _finish(new_seg(side), N)))
    return _merge_sequential(segs) if sequential else segs


def _merge_sequential(segs):
    """A launch that is ONLY the border-padded affine (kornia's warp_affine as its own resample) followed by a fused launch
    without an affine slot of its own, all on one image size, becomes ONE launch in the kernel's sequential form (`seq`: the
    homography slot interpolates an intermediate image whose integer pixels are the affine interpolation of the source,
    ffvc_augment_seq_fwd): the same values as the two launches, without the intermediate batch in memory.  The default set
    Af -> Pe -> Ji -> Er is one launch again, now with kornia's two interpolations."""
    out = []
    for kind, prm in segs:
        prev = out[-1] if out else None
        if (kind == "fused" and prev is not None and prev[0] == "fused" and prev[1].get("_affine_only") and not prm.get("_has_A") and
                prev[1]["src"] == prev[1]["out"] == prm["src"] == prm["out"] and not prev[1].get("seq")):
            merged = dict(prm)
            merged["ainv"] = prev[1]["ainv"]
            merged["seq"] = 1
            merged["_has_A"] = True
            merged["_affine_only"] = False
            out[-1] = ("fused", merged)
        else:
            out.append((kind, prm))
    return out


def _finish(seg, N):
    Hi = torch.linalg.inv(seg["H"])
    Hi = Hi / Hi[:, 2:3, 2:3]
    if seg["A"] is None:
        ainv = torch.tensor([1.0, 0, 0, 0, 1.0, 0], dtype=_F64).repeat(N, 1)
    else:
        Ai = torch.linalg.inv(seg["A"])
        ainv = Ai[:, :2, :].reshape(N, 6)
    eye = torch.eye(3, dtype=_F64).reshape(1, 9).repeat(N, 1)
    out = {"pinv": Hi.reshape(N, 9).float().contiguous(), "ainv": ainv.float().contiguous(), "cmat": eye.float().contiguous(),
           "coff": torch.zeros(N, 3), "erase": seg["erase"].contiguous(), "gn": seg["gn"].float().contiguous(),
           "src": seg["src"], "out": seg["out"],
           # (plan-level bookkeeping for _merge_sequential; not kernel parameters)
           "_has_A": seg["A"] is not None,
           "_affine_only": (seg["A"] is not None and seg["n_geo"] == 1 and seg["cj"] is None and not seg["has_erase"] and
                            not bool(seg["gn"].any()))}
    if seg["cj"] is not None:
        out["cj"] = seg["cj"].float().contiguous()
    return out


def draw_params(N, S, augs=DEFAULT, generator=None, p=0.7, src_size=None):
    """The single-launch form: -> dict of CPU tensors pinv (N,9), ainv (N,6), cmat (N,9), coff (N,3), cj (N,8, when the chain
    jitters), erase (N,4) i32, gn (N,).  Raises for chains that need more than one launch ('Sh', 'Et', 'Ts', or an order that
    leaves geometry -> colour -> erase): MakeCutouts runs those through plan()."""
    segs = plan(draw_chain(N, S, augs, generator, p, src_size), N, S, src_size, sequential=False)
    if len(segs) != 1:
        raise NotImplementedError(f"augs={list(augs)} needs {len(segs)} launches: use augment.plan() (MakeCutouts does)")
    prm = {k: v for k, v in segs[0][1].items() if not k.startswith("_")}
    prm.pop("src")
    prm.pop("out")
    return prm


def to_device(plan_or_params, device):
    """plan() segments (or one parameter dict) with every tensor on `device`."""
    mv = lambda d: {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in d.items()}   # noqa: E731
    if isinstance(plan_or_params, dict):
        return mv(plan_or_params)
    return [(kind, mv(prm)) for kind, prm in plan_or_params]
