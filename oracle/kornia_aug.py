"""Oracle (CPU) restatement of the kornia 0.5.10 augmentations that MakeCutouts chains (reference main.py:164-201).
TEST INFRASTRUCTURE ONLY — nothing under feed_forward_vqgan_clip_amd/ imports this module.

PARITY UNPINNED: kornia is neither in /root/reference nor installed in this image (requirements.txt pins kornia==0.5.10);
every function below restates the published 0.5.10 source of the function it names, INDEPENDENTLY of
feed_forward_vqgan_clip_amd/augment.py (which composes the chain into one resample).  What is restated:

  * each operator is applied to the output of the previous one — kornia's `nn.Sequential(*augment_list)` (main.py:199)
    resamples sequentially; an operator touches the samples its own Bernoulli(p) draw selects and passes the others on;
  * `warp_affine` / `warp_perspective` (kornia/geometry/transform/imgwarp.py) normalise the pixel homography with
    `normalize_homography` (pixel i <-> 2 i / (W - 1) - 1, the align_corners=True convention) but sample with
    `F.grid_sample(..., align_corners=False)` (the default of RandomAffine / RandomPerspective, kornia/augmentation/
    augmentation.py): the mismatch is reproduced as written, it is part of what the reference computes;
  * RandomAffine pads with 'border' (main.py:177), RandomPerspective with zeros (each of the four bilinear taps that falls outside
    the image counts as zero, so the image fades out over one pixel);
  * ColorJitter (kornia/augmentation/augmentation.py::ColorJitter.apply_transform) runs brightness -> contrast -> saturation ->
    hue in a random order `torch.randperm(4)`, saturation and hue through kornia's own rgb_to_hsv / hsv_to_rgb
    (kornia/color/hsv.py), brightness / contrast with clamps to [0, 1];
  * RandomErasing fills `xs .. xs + w - 1` x `ys .. ys + h - 1` with zeros; with same_on_batch=True one rectangle and ONE coin
    flip serve the whole batch;
  * RandomSharpness, RandomElasticTransform, RandomThinPlateSpline ('Sh', 'Et', 'Ts') as in kornia/enhance/adjust.py::sharpness,
    kornia/geometry/transform/elastic_transform.py::elastic_transform2d, kornia/geometry/transform/thin_plate_spline.py.

Samplers (`sample_*`) restate kornia/augmentation/random_generator/random_generator.py and `_range_bound`
(kornia/augmentation/utils/param_validation.py): a scalar `translate=0.1` becomes the range [-0.1, 0.1] clamped to the bounds
(0, 1) = [0, 0.1], read as (max_dx, max_dy) fractions -> NO horizontal shift, vertical shift U(-0.1 H, 0.1 H); the erasing aspect
ratio is a 50/50 mixture of U(r0, 1) and U(1, r1), not log-uniform.  All draws go through an explicit torch.Generator.
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------------------------
# geometry (kornia/geometry/transform/imgwarp.py, kornia/geometry/conversions.py)
# ---------------------------------------------------------------------------------------------------------------------
def normal_transform_pixel(h, w, dtype=torch.float64):
    """pixel -> [-1, 1] with pixel 0 at -1 and pixel w-1 at +1 (conversions.normal_transform_pixel)."""
    wd = 2.0 / max(w - 1, 1e-14)
    hd = 2.0 / max(h - 1, 1e-14)
    return torch.tensor([[wd, 0.0, -1.0], [0.0, hd, -1.0], [0.0, 0.0, 1.0]], dtype=dtype)


def normalize_homography(M, hw_src, hw_dst):
    """dst_norm <- src_norm version of the pixel homography M (N,3,3)."""
    src = normal_transform_pixel(*hw_src, dtype=M.dtype)
    dst = normal_transform_pixel(*hw_dst, dtype=M.dtype)
    return dst[None] @ M @ torch.linalg.inv(src)[None]


def warp_affine(img, M, padding_mode="border", align_corners=False):
    """imgwarp.warp_affine: M (N,2,3) or (N,3,3) maps SOURCE pixels to DESTINATION pixels."""
    N, C, H, W = img.shape
    M3 = torch.eye(3, dtype=torch.float64).repeat(N, 1, 1)
    M3[:, :2, :] = M[:, :2, :].to(torch.float64)
    inv = torch.linalg.inv(normalize_homography(M3, (H, W), (H, W)))
    grid = F.affine_grid(inv[:, :2, :].to(img.dtype), [N, C, H, W], align_corners=align_corners)
    return F.grid_sample(img, grid, mode="bilinear", padding_mode=padding_mode, align_corners=align_corners)


def warp_perspective(img, M, padding_mode="zeros", align_corners=False):
    """imgwarp.warp_perspective: the sampling grid is create_meshgrid(normalized_coordinates=True) = linspace(-1, 1, W)
    pushed through the inverse normalised homography."""
    N, C, H, W = img.shape
    inv = torch.linalg.inv(normalize_homography(M.to(torch.float64), (H, W), (H, W)))
    xs = torch.linspace(-1, 1, W, dtype=torch.float64)
    ys = torch.linspace(-1, 1, H, dtype=torch.float64)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    pts = torch.stack([gx, gy, torch.ones_like(gx)], dim=-1).view(1, H * W, 3)         # (1, HW, 3)
    q = pts @ inv.transpose(1, 2)                                                       # (N, HW, 3)
    z = q[..., 2:3]
    z = torch.where(z.abs() > 1e-8, z, torch.ones_like(z))                              # transform_points' eps guard
    grid = (q[..., :2] / z).view(N, H, W, 2).to(img.dtype)
    return F.grid_sample(img, grid, mode="bilinear", padding_mode=padding_mode, align_corners=align_corners)


def get_affine_matrix2d(angle_deg, translations, center, scale=None):
    """kornia get_affine_matrix2d without shear: get_rotation_matrix2d(center, -angle, scale), then += translations."""
    N = angle_deg.shape[0]
    a = torch.deg2rad(-angle_deg.to(torch.float64))
    sc = torch.ones(N, dtype=torch.float64) if scale is None else scale.to(torch.float64)
    # angle_to_rotation_matrix(a) = [[cos a, sin a], [-sin a, cos a]]
    alpha, beta = torch.cos(a) * sc, torch.sin(a) * sc
    x, y = center[:, 0].to(torch.float64), center[:, 1].to(torch.float64)
    M = torch.zeros(N, 3, 3, dtype=torch.float64)
    M[:, 0, 0], M[:, 0, 1], M[:, 0, 2] = alpha, beta, (1 - alpha) * x - beta * y
    M[:, 1, 0], M[:, 1, 1], M[:, 1, 2] = -beta, alpha, beta * x + (1 - alpha) * y
    M[:, 2, 2] = 1.0
    M[:, :2, 2] += translations.to(torch.float64)
    return M


def get_perspective_transform(src, dst):
    """(N,4,2) x (N,4,2) -> (N,3,3) with dst ~ H src (imgwarp.get_perspective_transform: the 8x8 DLT system)."""
    N = src.shape[0]
    src, dst = src.to(torch.float64), dst.to(torch.float64)
    x, y, u, v = src[..., 0], src[..., 1], dst[..., 0], dst[..., 1]
    z, o = torch.zeros_like(x), torch.ones_like(x)
    ax = torch.stack([x, y, o, z, z, z, -x * u, -y * u], dim=-1)
    ay = torch.stack([z, z, z, x, y, o, -x * v, -y * v], dim=-1)
    A = torch.stack([ax, ay], dim=2).view(N, 8, 8)
    b = torch.stack([u, v], dim=2).view(N, 8, 1)
    h = torch.linalg.solve(A, b).view(N, 8)
    return torch.cat([h, torch.ones(N, 1, dtype=torch.float64)], dim=1).view(N, 3, 3)


# ---------------------------------------------------------------------------------------------------------------------
# colour (kornia/color/hsv.py, kornia/enhance/adjust.py)
# ---------------------------------------------------------------------------------------------------------------------
def rgb_to_hsv(image, eps=1e-6):
    maxc, _ = image.max(-3)
    maxc_mask = image == maxc.unsqueeze(-3)
    _, max_indices = ((maxc_mask.cumsum(-3) == 1) & maxc_mask).max(-3)
    minc = image.min(-3)[0]
    v = maxc
    deltac = maxc - minc
    s = deltac / (v + eps)
    deltac = torch.where(deltac == 0, torch.ones_like(deltac), deltac)
    maxc_tmp = maxc.unsqueeze(-3) - image
    rc, gc, bc = maxc_tmp[..., 0, :, :], maxc_tmp[..., 1, :, :], maxc_tmp[..., 2, :, :]
    h = torch.stack([bc - gc, 2.0 * deltac + rc - bc, 4.0 * deltac + gc - rc], dim=-3)
    h = torch.gather(h, dim=-3, index=max_indices[..., None, :, :]).squeeze(-3)
    h = h / deltac
    h = (h / 6.0) % 1.0
    return torch.stack([2 * math.pi * h, s, v], dim=-3)


def hsv_to_rgb(image):
    h = image[..., 0, :, :] / (2 * math.pi)
    s, v = image[..., 1, :, :], image[..., 2, :, :]
    hi = torch.floor(h * 6) % 6
    f = ((h * 6) % 6) - hi
    one = torch.tensor(1.0, dtype=image.dtype)
    p, q, t = v * (one - s), v * (one - f * s), v * (one - (one - f) * s)
    hi = hi.long()
    indices = torch.stack([hi, hi + 6, hi + 12], dim=-3)
    out = torch.stack((v, q, p, p, t, v, t, v, v, q, p, p, p, p, t, v, v, q), dim=-3)
    return torch.gather(out, -3, indices)


def adjust_brightness(img, factor):
    return (img + factor.view(-1, 1, 1, 1)).clamp(0.0, 1.0)


def adjust_contrast(img, factor):
    return (img * factor.view(-1, 1, 1, 1)).clamp(0.0, 1.0)


def adjust_saturation(img, factor):
    hsv = rgb_to_hsv(img)
    h, s, v = hsv[:, 0], hsv[:, 1], hsv[:, 2]
    return hsv_to_rgb(torch.stack([h, (s * factor.view(-1, 1, 1)).clamp(0.0, 1.0), v], dim=1))


def adjust_hue(img, factor_rad):
    hsv = rgb_to_hsv(img)
    h, s, v = hsv[:, 0], hsv[:, 1], hsv[:, 2]
    return hsv_to_rgb(torch.stack([torch.fmod(h + factor_rad.view(-1, 1, 1), 2 * math.pi), s, v], dim=1))


def color_jitter(img, brightness_factor, contrast_factor, saturation_factor, hue_factor, order):
    """ColorJitter.apply_transform: the four adjustments in the batch-wide random `order` (a permutation of 0..3)."""
    ops = [lambda x: adjust_brightness(x, brightness_factor - 1),
           lambda x: adjust_contrast(x, contrast_factor),
           lambda x: adjust_saturation(x, saturation_factor),
           lambda x: adjust_hue(x, hue_factor * 2 * math.pi)]
    for i in [int(k) for k in order]:
        img = ops[i](img)
    return img


def sharpness(img, factor):
    """kornia.enhance.sharpness: blend of the 3x3-smoothed image (kernel [[1,1,1],[1,5,1],[1,1,1]] / 13, borders keep the
    original pixels) with the original: out = blur + (orig - blur) * factor, clamped to [0, 1] unless 0 < factor < 1."""
    N, C, H, W = img.shape
    k = torch.tensor([[1.0, 1.0, 1.0], [1.0, 5.0, 1.0], [1.0, 1.0, 1.0]], dtype=img.dtype).view(1, 1, 3, 3).repeat(C, 1, 1, 1) / 13
    deg = F.conv2d(img, k, bias=None, padding=0, stride=1, groups=C).clamp(0.0, 1.0)
    mask = F.pad(torch.ones_like(deg), [1, 1, 1, 1])
    res = torch.where(mask == 1, F.pad(deg, [1, 1, 1, 1]), img)
    f = factor.view(-1, 1, 1, 1).to(img.dtype)
    out = res + (img - res) * f
    inside = ((f > 0) & (f < 1)) | (f == 0) | (f == 1)          # _blend_one returns the unclamped blend for factors in [0, 1]
    return torch.where(inside, out, out.clamp(0.0, 1.0))


# ---------------------------------------------------------------------------------------------------------------------
# dense warps
# ---------------------------------------------------------------------------------------------------------------------
def _gaussian_kernel1d(ksize, sigma, dtype):
    x = torch.arange(ksize, dtype=dtype) - ksize // 2
    if ksize % 2 == 0:
        x = x + 0.5
    g = torch.exp(-x.pow(2) / (2 * sigma ** 2))
    return g / g.sum()


def elastic_transform2d(img, noise, kernel_size=(63, 63), sigma=(32.0, 32.0), alpha=(1.0, 1.0), align_corners=False):
    """kornia elastic_transform2d: noise (N,2,H,W) in [-1,1]; each component is blurred with a 63x63 Gaussian (sigma 32,
    'reflect' border as in filter2d), scaled by alpha, added to the identity grid in NORMALISED coordinates (meshgrid
    linspace(-1,1)) and sampled bilinearly with zero padding."""
    N, C, H, W = img.shape
    kx = _gaussian_kernel1d(kernel_size[1], sigma[1], img.dtype)
    ky = _gaussian_kernel1d(kernel_size[0], sigma[0], img.dtype)
    k2 = (ky[:, None] * kx[None, :]).view(1, 1, kernel_size[0], kernel_size[1])

    def blur(t):        # filter2d(border_type='reflect')
        ph, pw = kernel_size[0] // 2, kernel_size[1] // 2
        return F.conv2d(F.pad(t, [pw, pw, ph, ph], mode="reflect"), k2)

    disp_x = blur(noise[:, :1]) * alpha[0]
    disp_y = blur(noise[:, 1:]) * alpha[1]
    disp = torch.cat([disp_x, disp_y], dim=1).permute(0, 2, 3, 1)
    ys, xs = torch.linspace(-1, 1, H, dtype=img.dtype), torch.linspace(-1, 1, W, dtype=img.dtype)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    grid = (torch.stack([gx, gy], dim=-1)[None] + disp).clamp(-1, 1)
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=align_corners)


def _tps_kernel(d2):
    return 0.5 * d2 * torch.log(d2 + 1e-8)       # k(r) = r^2 log r = 0.5 r^2 log r^2 (_kernel_distance, eps 1e-8)


def get_tps_transform(points_src, points_dst):
    """thin_plate_spline.get_tps_transform: weights of the spline that maps points_src -> points_dst ((N,P,2) each).  As
    published, the kernel matrix holds the distances BETWEEN points_src and points_dst (`_pair_square_euclidean(points_src,
    points_dst)`): the kernel is centred on `points_dst`, which is what warp_image_tps is then given as `kernel_centers`."""
    N, P, _ = points_src.shape
    d2 = (points_src[:, :, None, :] - points_dst[:, None, :, :]).pow(2).sum(-1)
    k = _tps_kernel(d2)
    ones = torch.ones(N, P, 1, dtype=points_src.dtype)
    pmat = torch.cat([ones, points_src], dim=-1)                                   # (N,P,3)
    lmat = torch.cat([torch.cat([k, pmat], dim=-1),
                      torch.cat([pmat.transpose(1, 2), torch.zeros(N, 3, 3, dtype=points_src.dtype)], dim=-1)], dim=1)
    rhs = torch.cat([points_dst, torch.zeros(N, 3, 2, dtype=points_src.dtype)], dim=1)
    w = torch.linalg.solve(lmat, rhs)
    return w[:, :-3], w[:, -3:]                                                    # kernel weights (N,P,2), affine (N,3,2)


def warp_points_tps(points, kernel_centers, kernel_weights, affine_weights):
    d2 = (points[:, :, None, :] - kernel_centers[:, None, :, :]).pow(2).sum(-1)
    k = _tps_kernel(d2)
    return (k @ kernel_weights) + affine_weights[:, None, 0] + points @ affine_weights[:, 1:]


def thin_plate_spline(img, src, dst, align_corners=False):
    """RandomThinPlateSpline.apply_transform: `get_tps_transform(dst, src)` then `warp_image_tps(input, src, kernel, affine)`:
    the spline takes every destination grid point (normalised coordinates) to its source location."""
    N, C, H, W = img.shape
    kw, aw = get_tps_transform(dst.to(torch.float64), src.to(torch.float64))
    ys, xs = torch.linspace(-1, 1, H, dtype=torch.float64), torch.linspace(-1, 1, W, dtype=torch.float64)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    pts = torch.stack([gx, gy], dim=-1).view(1, H * W, 2).expand(N, H * W, 2)
    warped = warp_points_tps(pts, src.to(torch.float64), kw, aw).view(N, H, W, 2).to(img.dtype)
    return F.grid_sample(img, warped, mode="bilinear", padding_mode="zeros", align_corners=align_corners)


def erase_rectangles(img, xs, ys, widths, heights):
    N, C, H, W = img.shape
    xx = torch.arange(W).view(1, 1, W)
    yy = torch.arange(H).view(1, H, 1)
    m = (xx >= xs.view(N, 1, 1)) & (xx < (xs + widths).view(N, 1, 1)) & (yy >= ys.view(N, 1, 1)) & (yy < (ys + heights).view(N, 1, 1))
    return torch.where(m[:, None], torch.zeros_like(img), img)


# ---------------------------------------------------------------------------------------------------------------------
# samplers (kornia/augmentation/random_generator/random_generator.py) — explicit torch.Generator, float64 draws
# ---------------------------------------------------------------------------------------------------------------------
def _u(gen, n, lo, hi):
    return lo + (hi - lo) * torch.rand(n, generator=gen, dtype=torch.float64)


def _range_bound(factor, center=0.0, bounds=(0.0, float("inf"))):
    """param_validation._range_bound for a scalar factor: [center - f, center + f] clamped to `bounds`."""
    return (min(max(center - factor, bounds[0]), bounds[1]), min(max(center + factor, bounds[0]), bounds[1]))


def sample_chain(N, size, augs=("Af", "Pe", "Ji", "Er"), generator=None):
    """Raw parameters of every operator of the chain for a batch of N images of side `size`, in list order:
    [(name, dict)], each dict with `on` (N,) bool = the operator's Bernoulli(p) draw (batch-wide for same_on_batch)."""
    g = generator
    H = W = size
    out = []
    for a in augs:
        if a == "Af":       # RandomAffine(degrees=15, translate=0.1, p=0.7, padding_mode='border')   main.py:177
            lo, hi = _range_bound(0.1, 0.0, (0.0, 1.0))                 # translate=0.1 -> (0, 0.1) = (max_dx, max_dy) fractions
            max_dx, max_dy = lo * W, hi * H
            out.append((a, dict(on=torch.rand(N, generator=g, dtype=torch.float64) < 0.7, angle=_u(g, N, -15.0, 15.0),
                                translations=torch.stack([_u(g, N, -max_dx, max_dx), _u(g, N, -max_dy, max_dy)], dim=1),
                                center=torch.tensor([[(W - 1) / 2.0, (H - 1) / 2.0]], dtype=torch.float64).repeat(N, 1))))
        elif a == "Ro":     # RandomRotation(degrees=15, p=0.7)                                         main.py:175
            out.append((a, dict(on=torch.rand(N, generator=g, dtype=torch.float64) < 0.7, angle=_u(g, N, -15.0, 15.0),
                                translations=torch.zeros(N, 2, dtype=torch.float64),
                                center=torch.tensor([[(W - 1) / 2.0, (H - 1) / 2.0]], dtype=torch.float64).repeat(N, 1))))
        elif a == "Pe":     # RandomPerspective(distortion_scale=0.7, p=0.7)                            main.py:173
            start = torch.tensor([[0.0, 0.0], [W - 1.0, 0.0], [W - 1.0, H - 1.0], [0.0, H - 1.0]], dtype=torch.float64)
            sign = torch.tensor([[1.0, 1.0], [-1.0, 1.0], [-1.0, -1.0], [1.0, -1.0]], dtype=torch.float64)
            fac = torch.tensor([0.7 * W / 2, 0.7 * H / 2], dtype=torch.float64)
            rv = torch.rand(N, 4, 2, generator=g, dtype=torch.float64)
            out.append((a, dict(on=torch.rand(N, generator=g, dtype=torch.float64) < 0.7, start=start[None].repeat(N, 1, 1),
                                end=start[None] + fac * rv * sign[None])))
        elif a in ("Ji", "Ji2"):  # ColorJitter(hue=.1, saturation=.1, p=.7) / (brightness=.1, contrast=.1, saturation=.05, hue=.05, p=.5)
            br, ct, sa, hu, p = (0.0, 0.0, 0.1, 0.1, 0.7) if a == "Ji" else (0.1, 0.1, 0.05, 0.05, 0.5)
            b, c, s = _range_bound(br, 1.0, (0.0, 2.0)), _range_bound(ct, 1.0), _range_bound(sa, 1.0)
            out.append((a, dict(on=torch.rand(N, generator=g, dtype=torch.float64) < p, brightness=_u(g, N, *b), contrast=_u(g, N, *c),
                                saturation=_u(g, N, *s), hue=_u(g, N, -hu, hu), order=torch.randperm(4, generator=g))))
        elif a in ("Er", "Er2"):  # RandomErasing((.1,.4), (.3,1/.3), same_on_batch=(a == 'Er'), p=0.7)    main.py:185-187
            same = a == "Er"
            n = 1 if same else N
            area = _u(g, n, 0.1, 0.4) * H * W
            r1, r2 = _u(g, n, 0.3, 1.0), _u(g, n, 1.0, 1 / 0.3)
            ratio = torch.where(torch.rand(n, generator=g, dtype=torch.float64).round().bool(), r1, r2)
            hh = torch.sqrt(area * ratio).round().clamp(1, H)
            ww = torch.sqrt(area / ratio).round().clamp(1, W)
            xs = (_u(g, n, 0.0, 1.0) * (W - ww + 1)).floor()
            ys = (_u(g, n, 0.0, 1.0) * (H - hh + 1)).floor()
            on = torch.rand(n, generator=g, dtype=torch.float64) < 0.7
            ex = (lambda t: t.expand(N).clone()) if same else (lambda t: t)
            out.append((a, dict(on=ex(on), xs=ex(xs).long(), ys=ex(ys).long(), widths=ex(ww).long(), heights=ex(hh).long())))
        elif a == "Sh":     # RandomSharpness(sharpness=0.4, p=0.7): factor U(0.6, 1.4)                  main.py:169
            lo, hi = _range_bound(0.4, 1.0)
            out.append((a, dict(on=torch.rand(N, generator=g, dtype=torch.float64) < 0.7, factor=_u(g, N, lo, hi))))
        elif a == "Et":     # RandomElasticTransform(p=0.7): noise U(-1,1) per pixel and axis            main.py:179
            out.append((a, dict(on=torch.rand(N, generator=g, dtype=torch.float64) < 0.7,
                                noise=torch.rand(N, 2, H, W, generator=g, dtype=torch.float64) * 2 - 1)))
        elif a == "Ts":     # RandomThinPlateSpline(scale=0.3, p=0.7): 5 control points moved by U(-.3,.3) (normalised units)
            src = torch.tensor([[-1.0, -1.0], [-1.0, 1.0], [1.0, -1.0], [1.0, 1.0], [0.0, 0.0]], dtype=torch.float64)[None].repeat(N, 1, 1)
            out.append((a, dict(on=torch.rand(N, generator=g, dtype=torch.float64) < 0.7, src=src,
                                dst=src + _u(g, N * 10, -0.3, 0.3).view(N, 5, 2))))
        elif a == "Gn":     # RandomGaussianNoise(mean=0, std=1, p=0.5): the noise tensor itself is passed to apply_chain
            out.append((a, dict(on=torch.rand(N, generator=g, dtype=torch.float64) < 0.5)))
        else:
            raise NotImplementedError(f"kornia_aug.sample_chain: '{a}' (size-changing operators are restated in augment tests only)")
    return out


def apply_chain(batch, chain, gn_noise=None):
    """`nn.Sequential(*augment_list)(batch)` for the operators sample_chain knows: batch (N,3,S,S) fp32/64 in [0,1]."""
    x = batch
    for name, p in chain:
        on = p["on"]
        if not bool(on.any()):
            continue
        idx = on.nonzero().squeeze(1)
        sub = x[idx]
        if name == "Af":
            M = get_affine_matrix2d(p["angle"][idx], p["translations"][idx], p["center"][idx])
            sub = warp_affine(sub, M, padding_mode="border")
        elif name == "Ro":      # RandomRotation: zero padding and align_corners=True (its own default, unlike RandomAffine)
            M = get_affine_matrix2d(p["angle"][idx], p["translations"][idx], p["center"][idx])
            sub = warp_affine(sub, M, padding_mode="zeros", align_corners=True)
        elif name == "Pe":
            sub = warp_perspective(sub, get_perspective_transform(p["start"][idx], p["end"][idx]))
        elif name in ("Ji", "Ji2"):
            f = lambda k: p[k][idx].to(sub.dtype)    # noqa: E731
            sub = color_jitter(sub, f("brightness"), f("contrast"), f("saturation"), f("hue"), p["order"])
        elif name in ("Er", "Er2"):
            sub = erase_rectangles(sub, p["xs"][idx], p["ys"][idx], p["widths"][idx], p["heights"][idx])
        elif name == "Sh":
            sub = sharpness(sub, p["factor"][idx])
        elif name == "Et":
            sub = elastic_transform2d(sub, p["noise"][idx].to(sub.dtype))
        elif name == "Ts":
            sub = thin_plate_spline(sub, p["src"][idx], p["dst"][idx])
        elif name == "Gn":
            sub = sub + gn_noise[idx]
        x = x.index_copy(0, idx, sub.to(x.dtype))
    return x
