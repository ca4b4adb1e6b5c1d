"""Random parameters of the default MakeCutouts augmentations (main.py:164-165,172,178,182,190).

kornia 0.5.10 is not available offline, so its samplers are restated from their documented
distributions — statistically equivalent, parity unpinned:
  'Af' RandomAffine(degrees=15, translate=0.1, p=0.7, padding_mode='border'): angle ~ U(-15, 15) deg about the image
       centre, shift ~ U(-0.1, 0.1) * size per axis
  'Pe' RandomPerspective(distortion_scale=0.7, p=0.7): every corner moves inwards by U(0, 0.35 * size) per axis
  'Ji' ColorJitter(hue=0.1, saturation=0.1, p=0.7): hue shift U(-0.1, 0.1) turns, saturation U(0.9, 1.1); applied as a
       rotation / scaling of the chroma plane in YIQ (a linear RGB->RGB matrix)
  'Er' RandomErasing((.1,.4), (.3, 1/.3), same_on_batch=True, p=0.7): ONE rectangle (and one coin flip) per batch
Each augmentation is applied per sample with probability p.  Only tiny parameter tensors are produced here; the
resampling itself runs in ffvc_augment_fwd/bwd.
"""
import math

import torch

SUPPORTED = ("Af", "Pe", "Ji", "Er")
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


def draw_params(N, S, augs=SUPPORTED, generator=None, p=0.7):
    """-> dict of CPU tensors: pinv (N,9) f32, ainv (N,6) f32, cmat (N,9) f32, erase (N,4) i32."""
    for a in augs:
        if a not in SUPPORTED:
            raise NotImplementedError(f"augmentation '{a}' is not built on the HIP path (built: {SUPPORTED} and 'R')")
    g = generator
    rnd = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)  # noqa: E731
    c = (S - 1) / 2.0
    # --- affine ---------------------------------------------------------------------------------------------------
    ainv = torch.tensor([1.0, 0, 0, 0, 1.0, 0], dtype=torch.float64).repeat(N, 1)
    if "Af" in augs:
        on = rnd(N) < p
        th = torch.deg2rad((rnd(N) * 2 - 1) * 15.0)
        tx, ty = (rnd(N) * 2 - 1) * 0.1 * S, (rnd(N) * 2 - 1) * 0.1 * S
        cs, sn = torch.cos(th), torch.sin(th)
        a = torch.stack([cs, sn, c - cs * (c + tx) - sn * (c + ty), -sn, cs, c + sn * (c + tx) - cs * (c + ty)], dim=1)
        ainv = torch.where(on[:, None], a, ainv)
    # --- perspective ----------------------------------------------------------------------------------------------
    pinv = torch.eye(3, dtype=torch.float64).reshape(1, 9).repeat(N, 1)
    if "Pe" in augs:
        on = rnd(N) < p
        start = torch.tensor([[0.0, 0.0], [S - 1.0, 0.0], [S - 1.0, S - 1.0], [0.0, S - 1.0]], dtype=torch.float64)
        sign = torch.tensor([[1.0, 1.0], [-1.0, 1.0], [-1.0, -1.0], [1.0, -1.0]], dtype=torch.float64)
        end = start[None] + 0.7 * S / 2.0 * rnd(N, 4, 2) * sign[None]
        H = _homography(start[None].expand(N, 4, 2), end)
        Hi = torch.linalg.inv(H)
        Hi = Hi / Hi[:, 2:3, 2:3]
        pinv = torch.where(on[:, None], Hi.reshape(N, 9), pinv)
    # --- colour -----------------------------------------------------------------------------------------------------
    cmat = torch.eye(3, dtype=torch.float64).reshape(1, 9).repeat(N, 1)
    if "Ji" in augs:
        on = rnd(N) < p
        th = (rnd(N) * 2 - 1) * 0.1 * 2 * math.pi
        sat = 0.9 + 0.2 * rnd(N)
        rot = torch.zeros(N, 3, 3, dtype=torch.float64)
        rot[:, 0, 0] = 1.0
        rot[:, 1, 1] = sat * torch.cos(th)
        rot[:, 1, 2] = -sat * torch.sin(th)
        rot[:, 2, 1] = sat * torch.sin(th)
        rot[:, 2, 2] = sat * torch.cos(th)
        M = _YIQ_INV[None] @ rot @ _YIQ[None]
        cmat = torch.where(on[:, None], M.reshape(N, 9), cmat)
    # --- erasing (one rectangle for the whole batch) ----------------------------------------------------------------------
    erase = torch.zeros(N, 4, dtype=torch.int32)
    if "Er" in augs and float(rnd(1)) < p:
        area = (0.1 + 0.3 * float(rnd(1))) * S * S
        aspect = math.exp(math.log(0.3) + float(rnd(1)) * (math.log(1 / 0.3) - math.log(0.3)))
        h = max(1, min(S, int(round(math.sqrt(area * aspect)))))
        w = max(1, min(S, int(round(math.sqrt(area / aspect)))))
        x0 = int(float(rnd(1)) * (S - w + 1))
        y0 = int(float(rnd(1)) * (S - h + 1))
        erase[:] = torch.tensor([x0, y0, x0 + w, y0 + h], dtype=torch.int32)
    return {"pinv": pinv.float().contiguous(), "ainv": ainv.float().contiguous(), "cmat": cmat.float().contiguous(),
            "erase": erase.contiguous()}
