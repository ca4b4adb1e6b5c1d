"""Random parameters of the MakeCutouts augmentations (main.py:164-198).

kornia 0.5.10 is not available offline, so its samplers are restated from their documented distributions (SURVEY.md
App. A.4) — statistically equivalent, parity unpinned.  Every augmentation is applied per sample with its probability p;
only tiny parameter tensors are produced here, the resampling itself runs in ffvc_augment_fwd/bwd as ONE bilinear
resample  out(x) = C * in(A^-1(P^-1(x))) + c0  (zero outside P's source square, border clamp inside A), erase, + noise.

  geometric, composed in list order into the homography P (zero padding) — except a leading 'Af', which keeps kornia's
  border padding through the affine slot A:
    'Af'  RandomAffine(degrees=15, translate=0.1, p=0.7, padding_mode='border')   angle U(-15,15) deg, shift U(-.1,.1)*size
    'Pe'  RandomPerspective(distortion_scale=0.7, p=0.7)      every corner moves inwards by U(0, 0.35*size) per axis
    'Ro'  RandomRotation(degrees=15, p=0.7)                   angle U(-15,15) deg about the centre
    'Re'  RandomResizedCrop(scale=(0.1,1), ratio=(3/4,4/3), p=1)   crop of area U(.1,1)*S^2, log-uniform aspect, resized
    'Re2' RandomResizedCrop(scale=(0.9,1), ...)
    'R'   Resize(cut_size): bilinear, align_corners=False (main.py:145-152) — identity when the source already has cut_size
    'Cr'  RandomCrop(cut_size, p=0.5), 'Cc' CenterCrop(cut_size)   identities when the source already has cut_size; on a
          larger source (pool_size > cut_size or pool=False) a random / centred integer window (the crop is always taken:
          a batch cannot mix sizes)
  The chain tracks the current image side: it starts at `src_size` (pool_size, or the raw image side with pool=False) and
  becomes cut_size after 'R' / 'Re' / 'Re2' / 'Cr' / 'Cc' (out_size()).  Zero padding of
  'Pe' / 'Ro' is tested against the source frame (exact for resizes, approximate after a crop).
  colour (a 3x3 matrix + offset, composed in list order):
    'Ji'  ColorJitter(hue=0.1, saturation=0.1, p=0.7)         hue U(-.1,.1) turns, saturation U(.9,1.1), in the YIQ plane
    'Ji2' ColorJitter(brightness=.1, contrast=.1, saturation=.05, hue=.05, p=0.5)   brightness additive U(-.1,.1),
          contrast factor U(.9,1.1) (kornia's clamps to [0,1] between the steps are not applied)
  'Er'  RandomErasing((.1,.4), (.3,1/.3), same_on_batch=True, p=0.7)    ONE rectangle (and one coin flip) per batch
  'Er2' the same with same_on_batch=False                               one rectangle / coin flip per sample
  'Gn'  RandomGaussianNoise(std=1, p=0.5)                               per-sample N(0,1) noise, merged with MakeCutouts' own
                                                                        U(0,noise_fac)*N(0,1) term (sum of Gaussians)
'Sh' (sharpness), 'Et' (elastic), 'Ts' (thin-plate spline) need their own kernels and raise.
"""
import math

import torch

SUPPORTED = ("Af", "Pe", "Ji", "Er", "Ro", "Re", "Re2", "Cr", "Cc", "Ji2", "Er2", "Gn", "R")
RESIZING = ("R", "Re", "Re2", "Cr", "Cc")
DEFAULT = ("Af", "Pe", "Ji", "Er")
_YIQ = torch.tensor([[0.299, 0.587, 0.114], [0.5959, -0.2746, -0.3213], [0.2115, -0.5227, 0.3112]], dtype=torch.float64)
_YIQ_INV = torch.linalg.inv(_YIQ)


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


def _rot_about_centre(th, c):
    """(N,3,3) forward matrix of a rotation by th (radians) about (c, c)."""
    cs, sn = torch.cos(th), torch.sin(th)
    M = torch.zeros(th.shape[0], 3, 3, dtype=torch.float64)
    M[:, 0, 0], M[:, 0, 1], M[:, 0, 2] = cs, -sn, c - cs * c + sn * c
    M[:, 1, 0], M[:, 1, 1], M[:, 1, 2] = sn, cs, c - sn * c - cs * c
    M[:, 2, 2] = 1.0
    return M


def out_size(S, augs, src_size=None):
    """Side of the augmented batch: cut_size once the chain holds a resize / crop, else the source's side."""
    return S if (src_size is None or src_size == S or any(a in RESIZING for a in augs)) else src_size


def draw_params(N, S, augs=DEFAULT, generator=None, p=0.7, src_size=None):
    """-> dict of CPU tensors: pinv (N,9) f32, ainv (N,6) f32, cmat (N,9) f32, coff (N,3) f32, erase (N,4) i32,
    gn (N,) f32 (std of the extra per-sample Gaussian noise, 0 = none).  S = cut_size, src_size = side of the image the
    chain starts from (default S; the finished batch then has side out_size(S, augs, src_size)).  `p` overrides the 0.7 of
    the default set."""
    for a in augs:
        if a not in SUPPORTED:
            raise NotImplementedError(f"augmentation '{a}' is not built on the HIP path (built: {SUPPORTED} and 'R')")
    g = generator
    rnd = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)  # noqa: E731
    cut = S
    S = int(src_size or cut)                                          # current side of the image as the chain advances
    fin = out_size(cut, augs, S)                                       # side of the finished batch (erase rectangles live there)
    c = (S - 1) / 2.0
    eye3 = torch.eye(3, dtype=torch.float64)
    ainv = torch.tensor([1.0, 0, 0, 0, 1.0, 0], dtype=torch.float64).repeat(N, 1)
    Hfwd = eye3.reshape(1, 3, 3).repeat(N, 1, 1)                        # composite forward homography (source -> output)
    C = eye3.reshape(1, 3, 3).repeat(N, 1, 1)
    c0 = torch.zeros(N, 3, dtype=torch.float64)
    erase = torch.zeros(N, 4, dtype=torch.int32)
    gn = torch.zeros(N, dtype=torch.float64)
    first_geo = True

    def rect(n):
        """n erase rectangles (x0, y0, x1, y1): area U(.1,.4)*S^2, aspect log-uniform in (.3, 1/.3)."""
        area = (0.1 + 0.3 * rnd(n)) * fin * fin
        aspect = torch.exp(math.log(0.3) + rnd(n) * (math.log(1 / 0.3) - math.log(0.3)))
        h = torch.sqrt(area * aspect).round().clamp(1, fin)
        w = torch.sqrt(area / aspect).round().clamp(1, fin)
        x0 = (rnd(n) * (fin - w + 1)).floor()
        y0 = (rnd(n) * (fin - h + 1)).floor()
        return torch.stack([x0, y0, x0 + w, y0 + h], dim=1).to(torch.int32)

    for a in augs:
        if a == "Af":
            on = rnd(N) < p
            th = torch.deg2rad((rnd(N) * 2 - 1) * 15.0)
            tx, ty = (rnd(N) * 2 - 1) * 0.1 * S, (rnd(N) * 2 - 1) * 0.1 * S
            if first_geo:                                            # keeps kornia's border padding (inverse map, clamp)
                cs, sn = torch.cos(th), torch.sin(th)
                inv = torch.stack([cs, sn, c - cs * (c + tx) - sn * (c + ty), -sn, cs, c + sn * (c + tx) - cs * (c + ty)], dim=1)
                ainv = torch.where(on[:, None], inv, ainv)
            else:
                M = _rot_about_centre(th, c)
                M[:, 0, 2] += tx
                M[:, 1, 2] += ty
                Hfwd = torch.where(on[:, None, None], M @ Hfwd, Hfwd)
            first_geo = False
        elif a == "Pe":
            on = rnd(N) < p
            start = torch.tensor([[0.0, 0.0], [S - 1.0, 0.0], [S - 1.0, S - 1.0], [0.0, S - 1.0]], dtype=torch.float64)
            sign = torch.tensor([[1.0, 1.0], [-1.0, 1.0], [-1.0, -1.0], [1.0, -1.0]], dtype=torch.float64)
            end = start[None] + 0.7 * S / 2.0 * rnd(N, 4, 2) * sign[None]
            H = _homography(start[None].expand(N, 4, 2), end)
            Hfwd = torch.where(on[:, None, None], H @ Hfwd, Hfwd)
            first_geo = False
        elif a == "Ro":
            on = rnd(N) < 0.7
            M = _rot_about_centre(torch.deg2rad((rnd(N) * 2 - 1) * 15.0), c)
            Hfwd = torch.where(on[:, None, None], M @ Hfwd, Hfwd)
            first_geo = False
        elif a in ("Re", "Re2"):
            lo = 0.1 if a == "Re" else 0.9
            area = (lo + (1 - lo) * rnd(N)) * S * S
            ratio = torch.exp(math.log(0.75) + rnd(N) * (math.log(4 / 3) - math.log(0.75)))
            w = torch.sqrt(area * ratio).clamp(1, S)
            h = torch.sqrt(area / ratio).clamp(1, S)
            x0, y0 = rnd(N) * (S - w), rnd(N) * (S - h)
            M = torch.zeros(N, 3, 3, dtype=torch.float64)            # crop [x0, x0+w-1] x [y0, y0+h-1] -> [0, cut-1]^2
            M[:, 0, 0] = (cut - 1) / (w - 1).clamp_min(1e-6)
            M[:, 1, 1] = (cut - 1) / (h - 1).clamp_min(1e-6)
            M[:, 0, 2] = -x0 * M[:, 0, 0]
            M[:, 1, 2] = -y0 * M[:, 1, 1]
            M[:, 2, 2] = 1.0
            Hfwd = M @ Hfwd
            first_geo = False
            S, c = cut, (cut - 1) / 2.0
        elif a == "R":
            if S != cut:                                             # x_out = (x_in + .5) * cut / S - .5  (align_corners=False)
                M = torch.zeros(N, 3, 3, dtype=torch.float64)
                M[:, 0, 0] = M[:, 1, 1] = cut / S
                M[:, 0, 2] = M[:, 1, 2] = 0.5 * cut / S - 0.5
                M[:, 2, 2] = 1.0
                Hfwd = M @ Hfwd
                first_geo = False
                S, c = cut, (cut - 1) / 2.0
        elif a in ("Cr", "Cc"):
            if S < cut:
                raise ValueError(f"'{a}': the image ({S}) is smaller than cut_size ({cut})")
            if S > cut:                                              # integer window of the current image
                if a == "Cr":
                    x0 = (rnd(N) * (S - cut + 1)).floor()
                    y0 = (rnd(N) * (S - cut + 1)).floor()
                else:
                    x0 = y0 = torch.full((N,), float((S - cut) // 2), dtype=torch.float64)
                M = eye3.reshape(1, 3, 3).repeat(N, 1, 1)
                M[:, 0, 2], M[:, 1, 2] = -x0, -y0
                Hfwd = M @ Hfwd
                first_geo = False
                S, c = cut, (cut - 1) / 2.0
        elif a in ("Ji", "Ji2"):
            hue, sat, pj = (0.1, 0.1, p) if a == "Ji" else (0.05, 0.05, 0.5)
            on = rnd(N) < pj
            bright = (rnd(N) * 2 - 1) * 0.1 if a == "Ji2" else torch.zeros(N, dtype=torch.float64)
            contrast = 0.9 + 0.2 * rnd(N) if a == "Ji2" else torch.ones(N, dtype=torch.float64)
            th = (rnd(N) * 2 - 1) * hue * 2 * math.pi
            s_ = 1.0 - sat + 2 * sat * rnd(N)
            rot = torch.zeros(N, 3, 3, dtype=torch.float64)
            rot[:, 0, 0] = 1.0
            rot[:, 1, 1] = s_ * torch.cos(th)
            rot[:, 1, 2] = -s_ * torch.sin(th)
            rot[:, 2, 1] = s_ * torch.sin(th)
            rot[:, 2, 2] = s_ * torch.cos(th)
            M = (_YIQ_INV[None] @ rot @ _YIQ[None]) * contrast[:, None, None]     # brightness, contrast, then chroma
            off = (M @ bright[:, None, None].expand(N, 3, 1)).squeeze(-1)
            C = torch.where(on[:, None, None], M @ C, C)
            c0 = torch.where(on[:, None], (M @ c0[:, :, None]).squeeze(-1) + off, c0)
        elif a == "Er":
            if float(rnd(1)) < p:
                erase[:] = rect(1)[0]
        elif a == "Er2":
            on = rnd(N) < 0.7
            erase = torch.where(on[:, None], rect(N), erase)
        elif a == "Gn":
            gn = torch.where(rnd(N) < 0.5, torch.ones(N, dtype=torch.float64), gn)
    Hi = torch.linalg.inv(Hfwd)
    Hi = Hi / Hi[:, 2:3, 2:3]
    return {"pinv": Hi.reshape(N, 9).float().contiguous(), "ainv": ainv.float().contiguous(),
            "cmat": C.reshape(N, 9).float().contiguous(), "coff": c0.float().contiguous(), "erase": erase.contiguous(),
            "gn": gn.float().contiguous()}
