"""Gravitational lensing of a background image -- MI355X backend.

Same public surface as the reference's image_lens.py (function names, arguments,
return values, CLI flags), so `import image_lens` / `python image_lens.py --a 0.9`
keep working.  Coordinates are (y, x); FOV pairs are (horizontal, vertical).

Where the work happens:

  * every per-pixel array is produced on the GPU: `build_alpha_lookup` (lt_pixel_angles),
    `precompute_final_alpha_lookup[_2d]` (the metric's `trace_rays_batch`, i.e.
    lt_trace_batch_*), `render_lensed_image` (lt_shade);
  * `render_frame` / `main()` use the fused path (lt_render): pixel -> ray -> colour in one
    call, nothing per-ray crosses PCIe on the way in;  `main(staged=True)` runs the
    reference's three-call sequence instead, stage by stage on the GPU.

The scalar camera helpers (`_psi_frame`, `pixel_to_angles`, `angles_to_pixel`) are a few
float64 operations and stay on the host.  No CPU tracer exists in this package.
"""
from time import perf_counter

import json
import os

import numpy as np

import ltrace
from metrics import Kerr, Schwarzschild

WINDING_DTYPE = np.uint16
WINDING_MAX = np.iinfo(WINDING_DTYPE).max
Y_AXIS_REFINE_FRAC = 0.07
TRACE_CHUNK = 4_000_000   # rays per trace_rays_batch call (the reference uses 50 000 for its tqdm bar)


# ---------------------------------------------------------------------------------------------
# camera model (host scalars)
# ---------------------------------------------------------------------------------------------
def _psi_to_bh_direction(psi):
    """psi = (pitch_up, yaw_right) [rad] -> unit vector to the BH in camera axes (+x right, +y down,
    +z forward)."""
    pitch, yaw = psi
    return np.array([np.sin(yaw) * np.cos(pitch), -np.sin(pitch), np.cos(yaw) * np.cos(pitch)], dtype=np.float64)


def _unit(v, fallback):
    n = np.linalg.norm(v)
    if n < 1e-12:
        v = fallback()
        n = np.linalg.norm(v)
    return v / max(n, 1e-12)


def _psi_frame(psi):
    """(d, e_x, e_y, in_front): BH direction and the screen basis around it (Gram-Schmidt of the
    camera's x and y axes against d; e_x / e_y coincide with the image axes at psi = 0)."""
    d = _psi_to_bh_direction(psi)
    x_hat = np.array([1.0, 0.0, 0.0])
    y_hat = np.array([0.0, 1.0, 0.0])
    e_x = _unit(x_hat - (x_hat @ d) * d, lambda: y_hat - (y_hat @ d) * d)
    e_y = _unit(y_hat - (y_hat @ d) * d - (y_hat @ e_x) * e_x, lambda: np.cross(d, e_x))
    return d, e_x, e_y, bool(d[2] > 1e-12)


def _psi_to_cam_projection(psi):
    """(y_cam, x_cam, in_front) of the BH on the pinhole plane; NaNs if it is behind the camera."""
    d, _, _, front = _psi_frame(psi)
    if not front:
        return (np.nan, np.nan, False)
    return (float(d[1] / d[2]), float(d[0] / d[2]), True)


def _focal(image_dimension, fov):
    height, width = image_dimension
    return (width / 2) / np.tan(fov[0] / 2), (height / 2) / np.tan(fov[1] / 2)


def pixel_to_angles(pixel, image_dimension, fov, psi=(0.0, 0.0)):
    """(alpha, theta) of one pixel (y, x): angle from the BH direction and screen azimuth."""
    height, width = image_dimension
    fx, fy = _focal(image_dimension, fov)
    ray = np.array([(pixel[1] - width / 2) / fx, (pixel[0] - height / 2) / fy, 1.0])
    ray /= np.linalg.norm(ray)
    d, e_x, e_y, _ = _psi_frame(psi)
    return (float(np.arccos(np.clip(ray @ d, -1.0, 1.0))), float(np.arctan2(ray @ e_x, ray @ e_y)))


def angles_to_pixel(angles, image_dimension, fov, clip=False, psi=(0.0, 0.0)):
    """Inverse of pixel_to_angles -> (py, px); (-1, -1) (or (0, 0) with clip) behind the camera."""
    alpha, theta = angles
    height, width = image_dimension
    fx, fy = _focal(image_dimension, fov)
    d, e_x, e_y, _ = _psi_frame(psi)
    ray = np.cos(alpha) * d + np.sin(alpha) * (np.sin(theta) * e_x + np.cos(theta) * e_y)
    if ray[2] <= 1e-12:
        return (0, 0) if clip else (-1, -1)
    px = int(np.rint(ray[0] / ray[2] * fx + width / 2))
    py = int(np.rint(ray[1] / ray[2] * fy + height / 2))
    if clip:
        px, py = int(np.clip(px, 0, width - 1)), int(np.clip(py, 0, height - 1))
    return (py, px)


def _camera(image_dimension, fov, psi, r_obs=50.0, theta_obs=np.pi / 2):
    height, width = image_dimension
    return ltrace.Camera(int(width), int(height), float(fov[0]), float(fov[1]), float(psi[0]), float(psi[1]),
                         float(r_obs), float(theta_obs))


# ---------------------------------------------------------------------------------------------
# stage 1: per-pixel alpha
# ---------------------------------------------------------------------------------------------
def build_alpha_lookup(image_dimension, fov, decimals=None, psi=(0.0, 0.0)):
    """(H, W) float32 viewing angle per pixel corner (GPU: lt_pixel_angles)."""
    alpha, _, _ = ltrace.pixel_angles(_camera(image_dimension, fov, psi), want_theta=False)
    if decimals is not None:
        alpha = np.round(alpha.astype(np.float64), decimals).astype(np.float32)
    return alpha


# ---------------------------------------------------------------------------------------------
# stage 2: trace one ray per pixel
# ---------------------------------------------------------------------------------------------
def _empty_lookup(shape):
    return np.full(shape, np.nan, dtype=np.float32), np.zeros(shape, dtype=WINDING_DTYPE)


def precompute_final_alpha_lookup(alpha_lookup, alpha_crit, r_obs, metric, dedup=False):
    """Spherically symmetric metrics: every pixel traced from its alpha alone.
    -> (final_alpha f32, winding u16, total_rays, traced_rays).

    dedup=True traces every DISTINCT float32 alpha once and hands the result to all pixels that share it (the idea of
    the reference's dormant debugging_image_lense.py:634): a pinhole frame has 8-fold symmetry, so at most
    ~N/8 + O(sqrt N) values are distinct.  Same alpha, same ray: the lookups are byte-identical to the full trace; only
    `traced_rays` differs.  Off by default, as in the reference's live function (which traces all N)."""
    alpha = alpha_lookup.ravel().astype(np.float64)
    n = alpha.size
    if n == 0:
        fa, w = _empty_lookup(alpha_lookup.shape)
        return fa, w, n, 0
    inverse = None
    if dedup:
        alpha, inverse = np.unique(alpha, return_inverse=True)
    m = alpha.size
    fa = np.full(m, np.nan, dtype=np.float64)
    w = np.zeros(m, dtype=np.int64)
    for lo in range(0, m, TRACE_CHUNK):
        hi = min(lo + TRACE_CHUNK, m)
        metric.trace_rays_batch(r_obs, alpha[lo:hi], fa[lo:hi], w[lo:hi])
    if inverse is not None:
        fa, w = fa[inverse.ravel()], w[inverse.ravel()]
    return (fa.astype(np.float32).reshape(alpha_lookup.shape),
            np.clip(w, 0, WINDING_MAX).astype(WINDING_DTYPE).reshape(alpha_lookup.shape), n, m)


def precompute_final_alpha_lookup_2d(alpha_lookup, fov, alpha_crit, r_obs, metric,
                                     theta_obs=np.pi / 2, psi=(0.0, 0.0)):
    """Non-spherical metrics: (alpha, theta) per pixel, axis-refine columns, and the reference's
    top/bottom mirror (rows 0..(H+1)//2-1 traced, row j copied to row H-1-j) when theta_obs = pi/2
    and psi_y = 0.  -> (final_alpha f32, winding u16, total_rays, traced_rays)."""
    height, width = alpha_lookup.shape
    _, theta, refine_cols = ltrace.pixel_angles(_camera((height, width), fov, psi), Y_AXIS_REFINE_FRAC)
    mirror = bool(np.isclose(theta_obs, np.pi / 2) and np.isclose(psi[0], 0.0))
    rows = (height + 1) // 2 if mirror else height
    n = rows * width
    print(f"  tracing {n:,} rays " + ("with top/bottom symmetry " if mirror else "")
          + f"({alpha_lookup.size:,} pixels total)")
    fa_out, w_out = _empty_lookup((height, width))
    if n:
        alpha = alpha_lookup[:rows].ravel().astype(np.float64)
        th = np.ascontiguousarray(theta[:rows]).ravel()
        refine = np.broadcast_to(refine_cols[None, :], (rows, width)).ravel()
        fa = np.full(n, np.nan, dtype=np.float64)
        w = np.zeros(n, dtype=np.int64)
        for lo in range(0, n, TRACE_CHUNK):
            hi = min(lo + TRACE_CHUNK, n)
            metric.trace_rays_batch(r_obs, alpha[lo:hi], th[lo:hi], theta_obs,
                                    np.ascontiguousarray(refine[lo:hi]), fa[lo:hi], w[lo:hi])
        fa_out[:rows] = fa.astype(np.float32).reshape(rows, width)
        w_out[:rows] = np.clip(w, 0, WINDING_MAX).astype(WINDING_DTYPE).reshape(rows, width)
    if mirror and height // 2 > 0:
        half = height // 2
        fa_out[height - half:] = fa_out[:half][::-1]
        w_out[height - half:] = w_out[:half][::-1]
    return fa_out, w_out, int(alpha_lookup.size), int(n)


# ---------------------------------------------------------------------------------------------
# stage 3: colouring
# ---------------------------------------------------------------------------------------------
WINDING_COLORS = np.array([
    [0.0, 0.2, 1.0],   # blue
    [0.0, 0.7, 1.0],   # sky blue
    [0.0, 1.0, 0.4],   # green
    [1.0, 1.0, 0.0],   # yellow
    [1.0, 0.4, 0.0],   # orange
], dtype=np.float32)


def render_lensed_image(source_image, alpha_lookup, final_alpha_lookup, winding_lookup, alpha_crit, fov,
                        render_loop_around=False, psi=(0.0, 0.0)):
    """Output image from the lookups (GPU: lt_shade): captured -> black, exit angle beyond pi/2 ->
    WINDING_COLORS[min(winding, 4)], otherwise the background pixel the exit direction points at
    (magenta outside the frame, or wrapped with render_loop_around).  `alpha_lookup` and `alpha_crit`
    are accepted for signature compatibility; like the reference, nothing reads them."""
    src = np.asarray(source_image)
    out = ltrace.shade(_camera(src.shape[:2], fov, psi), src.astype(np.float32, copy=False),
                       final_alpha_lookup, winding_lookup, loop_around=render_loop_around)
    return out.astype(src.dtype, copy=False) if src.dtype.kind == "f" else out


def render_frame(source_image, metric, r_obs, fov, psi=(0.0, 0.0), theta_obs=np.pi / 2, integrator=None,
                 precision=None, schedule=None, tb_symmetry=False, render_loop_around=False,
                 want=("fa", "winding", "rgb"), gpus=1, devices=None):
    """Fused path (lt_render): all three stages in one GPU call.  source_image None -> shadow render
    (escaped = white).  Returns dict with 'fa', 'winding', 'rgb', ... and 'stats'.
    gpus > 1: the frame's rows are split block-cyclically over that many devices of this node
    (lt_render_multi; the reference's top/bottom mirror needs the whole frame on one device)."""
    if source_image is None:
        raise ValueError("render_frame needs a background; for a shadow use black_hole_shadow.render_traced")
    source_image = np.asarray(source_image)
    if source_image.ndim == 3 and source_image.shape[2] == 4:
        # RGBA input (a PNG background): the reference's renderer cannot colour winding pixels of a 4-channel image
        # (it assigns 3-channel WINDING_COLORS, image_lens.py:329-331) and turns black / magenta pixels transparent;
        # the colour planes are what is lensed here
        source_image = source_image[..., :3]
    shape = source_image.shape[:2]
    kerr = not metric.is_spherically_symmetric
    met = ltrace.Metric(ltrace.METRIC_KERR if kerr else ltrace.METRIC_SCHWARZSCHILD, 0, float(metric.M),
                        float(getattr(metric, "a", 0.0)))
    opts = ltrace.default_opts(
        integrator=integrator or getattr(metric, "integrator", "rk4"),
        precision=precision or getattr(metric, "precision", 32),
        schedule=schedule or getattr(metric, "schedule", "direct"),
        tb_symmetry=int(bool(tb_symmetry)), loop_around=int(bool(render_loop_around)),
        axis_refine_frac=Y_AXIS_REFINE_FRAC)
    cam = _camera(shape, fov, psi, r_obs, theta_obs)
    if gpus and gpus > 1:
        if tb_symmetry:
            raise ValueError("tb_symmetry (the reference's top/bottom mirror) needs gpus == 1")
        if devices is None and os.environ.get("LT_MULTI_DEVICES"):   # e.g. "0,0": rehearse 2 partitions on one GPU
            devices = [int(x) for x in os.environ["LT_MULTI_DEVICES"].split(",")]
        return ltrace.render_multi(cam, met, opts, gpus, devices=devices, background=source_image, want=want)
    return ltrace.render(cam, met, opts, background=source_image, want=want)


# ---------------------------------------------------------------------------------------------
# benchmark print, CLI
# ---------------------------------------------------------------------------------------------
def print_benchmark_summary(image_dimension, alpha_crit, total_rays, traced_rays, timings):
    height, width = image_dimension
    pixels = width * height
    render_t = max(timings.get("render", 0.0), 1e-12)
    total_t = max(timings.get("total", 0.0), 1e-12)
    print("\nBenchmark summary")
    print(f"  resolution: {width}x{height} ({pixels:,} pixels)")
    print(f"  alpha_crit: {alpha_crit:.6f} rad")
    print(f"  total rays: {total_rays:,}")
    print(f"  traced rays: {traced_rays:,}")
    for key in ("load_image", "build_lookup", "precompute", "render", "save_image", "total"):
        print(f"  {key:<26}{timings.get(key, 0.0):>10.3f} s")
    print(f"  {'render_throughput':<26}{(pixels / render_t) / 1e6:>10.2f} MPix/s")
    print(f"  {'overall_throughput':<26}{(pixels / total_t) / 1e6:>10.2f} MPix/s")
    if "gpu_integrate_ms" in timings:
        print(f"  {'gpu integrate kernel':<26}{timings['gpu_integrate_ms']:>10.3f} ms "
              f"({traced_rays / timings['gpu_integrate_ms'] / 1e3:.1f} Mrays/s)")


def synthetic_background(height, width, seed=0):
    """Seeded RGB texture in [0, 1] (uint8 / 255 like an imread of a JPEG): the reference ships no image."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(height, width, 3), dtype=np.uint8).astype(np.float32) / 255.0


def write_png_rgba8(path, rgba, level=1):
    """Write an (H, W, 4) uint8 array as a PNG (8-bit RGBA, filter 0, one IDAT) with zlib alone.
    mpimg.imsave (image_lens.py:510 of the reference) spends its time converting float RGB to RGBA8 and
    deflating at level 6; here the GPU epilogue already produced the RGBA8 bytes (same truncation,
    bit-exact) and level 1 is enough for a render that is written once.  Decodes to the same pixels."""
    import struct
    import zlib
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    if rgba.ndim != 3 or rgba.shape[2] != 4:
        raise ValueError("rgba must be (H, W, 4) uint8")
    h, w = rgba.shape[:2]
    raw = np.empty((h, 1 + 4 * w), dtype=np.uint8)
    raw[:, 0] = 0                       # filter type None on every scanline
    raw[:, 1:] = rgba.reshape(h, 4 * w)

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw.tobytes(), level)))
        f.write(chunk(b"IEND", b""))


def _lookup_cache_key(metric, r_obs, shape, fov, psi, mirror):
    """Everything the lookups depend on: metric and its backend knobs, observer, camera, the mirror quirk, and the build
    of the library that traced them (a new kernel build never reuses an old cache)."""
    return json.dumps({"metric": type(metric).__name__, "M": float(metric.M), "a": float(getattr(metric, "a", 0.0)),
                       "integrator": getattr(metric, "integrator", None), "precision": int(getattr(metric, "precision", 64)),
                       "r_obs": float(r_obs), "shape": [int(shape[0]), int(shape[1])], "fov": [float(fov[0]), float(fov[1])],
                       "psi": [float(psi[0]), float(psi[1])], "mirror": bool(mirror), "build": ltrace.build_id()}, sort_keys=True)


def load_lookup_cache(path, key):
    """(final_alpha, winding) from `path` if it was written for `key`, else None.  The file is ours (np.savez of two arrays
    and a JSON string): loaded without pickle."""
    try:
        with np.load(path, allow_pickle=False) as z:
            if str(z["key"]) == key:
                return z["final_alpha"].copy(), z["winding"].copy()
    except (OSError, KeyError, ValueError):
        pass
    return None


def save_lookup_cache(path, key, final_alpha, winding):
    with open(path, "wb") as f:          # (a file object: np.savez would append ".npz" to a bare name, and the next load would miss it)
        np.savez(f, key=np.array(key), final_alpha=final_alpha, winding=winding)


def main(metric=None, M=1.0, a=0.0, r_obs_mult=100.0, psi=(0.0, 0.0), vertical_fov_deg=40.0,
         image_path="image.jpg", output_path="lensed_image.png", synthetic=None, staged=False,
         integrator=None, precision=None, schedule=None, gpus=1, full_trace=False, dedup_alpha=False,
         lookup_cache=None):
    """`lookup_cache`: path of an .npz (the reference's .gitignore names `lookup_cache.npz`, it never wrote one): the
    final_alpha / winding lookups of this metric, observer and camera are stored there and reused by the next call with
    the same settings -- a new background then costs one colouring pass (lt_shade) instead of a trace.
    `dedup_alpha`: staged path, spherically symmetric metrics: trace distinct alphas only (precompute_final_alpha_lookup)."""
    import matplotlib.image as mpimg

    if metric is None:
        metric = (Schwarzschild(M=M, precision=precision) if a == 0
                  else Kerr(M=M, a=a, integrator=integrator, precision=precision, schedule=schedule))
    print(f"Metric: {type(metric).__name__} (M={metric.M}, a={getattr(metric, 'a', 0)})")
    timings = {}
    t_total = perf_counter()

    t0 = perf_counter()
    if synthetic:
        img = synthetic_background(int(synthetic[1]), int(synthetic[0]))
    else:
        img = mpimg.imread(image_path)
        if img.dtype == np.uint8:
            img = img.astype(np.float32) / 255.0
        if img.ndim == 3 and img.shape[2] == 4:
            print("Background has an alpha channel: lensing its RGB planes (see render_frame)")
            img = np.ascontiguousarray(img[..., :3])
    timings["load_image"] = perf_counter() - t0
    height, width = img.shape[:2]
    print(f"Image: {width}x{height}")

    r_obs = r_obs_mult * metric.M
    alpha_crit = metric.alpha_crit(r_obs)
    print(f"r_obs = {r_obs:.1f} M, alpha_crit = {np.degrees(alpha_crit):.4f} deg")
    vfov = np.radians(vertical_fov_deg)
    fov = (2 * np.arctan(np.tan(vfov / 2) * width / height), vfov)
    bh_y, bh_x, front = _psi_to_cam_projection(psi)
    where = ("behind observer" if not front else
             "inside FOV" if abs(bh_y) <= np.tan(fov[1] / 2) and abs(bh_x) <= np.tan(fov[0] / 2) else "outside FOV")
    print(f"BH screen offset: psi_y={np.degrees(psi[0]):.4f} deg, psi_x={np.degrees(psi[1]):.4f} deg ({where})")

    rgba8 = None
    mirror = (not full_trace) and gpus <= 1 and not metric.is_spherically_symmetric and abs(psi[0]) <= 1e-8
    # (the staged path mirrors whenever the reference does, the fused one unless --full-trace / several GPUs)
    mirrored = (not metric.is_spherically_symmetric and abs(psi[0]) <= 1e-8) if staged else mirror
    cache_key = _lookup_cache_key(metric, r_obs, (height, width), fov, psi, mirrored) if lookup_cache else None
    cached = load_lookup_cache(lookup_cache, cache_key) if lookup_cache else None
    if cached is not None:
        print(f"Lookup cache hit ({lookup_cache}): colouring only")
        fa, wd = cached
        t0 = perf_counter()
        lensed = render_lensed_image(img, None, fa, wd, alpha_crit, fov, False, psi=psi)
        timings["render"] = perf_counter() - t0
        total, traced = height * width, 0
    elif staged:
        print("Building per-pixel " + ("alpha" if metric.is_spherically_symmetric else "(alpha, theta)") + " lookup...")
        t0 = perf_counter()
        alpha_lookup = build_alpha_lookup((height, width), fov, psi=psi)
        timings["build_lookup"] = perf_counter() - t0
        t0 = perf_counter()
        if metric.is_spherically_symmetric:
            fa, wd, total, traced = precompute_final_alpha_lookup(alpha_lookup, alpha_crit, r_obs, metric, dedup=dedup_alpha)
        else:
            fa, wd, total, traced = precompute_final_alpha_lookup_2d(alpha_lookup, fov, alpha_crit, r_obs, metric, psi=psi)
        timings["precompute"] = perf_counter() - t0
        t0 = perf_counter()
        lensed = render_lensed_image(img, alpha_lookup, fa, wd, alpha_crit, fov, False, psi=psi)
        timings["render"] = perf_counter() - t0
        if lookup_cache:
            save_lookup_cache(lookup_cache, cache_key, fa, wd)
    else:
        # The reference traces the top half of the frame and mirrors it whenever the observer is equatorial and the
        # hole is not offset vertically (image_lens.py:218-220, :272-276 -- off by one row, quirk Q1).  Default: do
        # as the reference does, so that `python image_lens.py --a 0.9` gives the reference's picture; --full-trace
        # (and every multi-GPU render) traces every row instead.
        print(f"Fused GPU render (pixel -> ray -> colour) on {max(gpus, 1)} GPU(s); rows: "
              + ("top half traced, bottom half mirrored as in the reference" if mirror else "every row traced"))
        t0 = perf_counter()
        out = render_frame(img, metric, r_obs, fov, psi=psi, tb_symmetry=mirror,
                           want=("rgb", "rgba") + (("fa", "winding") if lookup_cache else ()), gpus=gpus)
        timings["render"] = perf_counter() - t0
        if lookup_cache:
            save_lookup_cache(lookup_cache, cache_key, np.asarray(out["fa"]), np.asarray(out["winding"]))
        timings["gpu_integrate_ms"] = out["stats"]["integrate_ms"]
        lensed, total, traced = out["rgb"], height * width, out["stats"]["rays"]   # mirrored rows are copies, not rays
        rgba8 = out["rgba"] if lensed.ndim == 3 else None

    t0 = perf_counter()
    if rgba8 is not None and str(output_path).lower().endswith(".png"):
        write_png_rgba8(output_path, rgba8)   # the epilogue kernel's RGBA8 == imsave's conversion of `lensed`
    else:
        mpimg.imsave(output_path, lensed)
    timings["save_image"] = perf_counter() - t0
    timings["total"] = perf_counter() - t_total
    print_benchmark_summary((height, width), alpha_crit, total, traced, timings)
    return lensed


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=float, default=1.0, help="BH mass")
    ap.add_argument("--a", type=float, default=0.0, help="BH spin (|a| <= M, 0 = Schwarzschild)")
    ap.add_argument("--r-obs", type=float, default=100.0, help="Observer distance in units of M (default: 100)")
    ap.add_argument("--psi-y", type=float, default=0.0, help="BH vertical offset in deg (+ = top, - = bottom)")
    ap.add_argument("--psi-x", type=float, default=0.0, help="BH horizontal offset in deg (+ = right, - = left)")
    ap.add_argument("--fov-v", type=float, default=40.0, help="Vertical field of view in deg")
    ap.add_argument("--lookup-cache", default=None, help="path of an .npz holding the final_alpha / winding lookups: written after a "
                                                         "trace, reused by the next call with the same metric, observer and camera")
    ap.add_argument("--dedup-alpha", action="store_true", help="--staged, a = 0: trace every distinct alpha once")
    # backend additions
    ap.add_argument("--image", default="image.jpg", help="background image (default: image.jpg)")
    ap.add_argument("--output", default="lensed_image.png")
    ap.add_argument("--synthetic", type=int, nargs=2, metavar=("W", "H"), help="use a seeded synthetic background")
    ap.add_argument("--staged", action="store_true", help="run the reference's three stages one by one")
    ap.add_argument("--integrator", choices=["rk4", "dp45", "dp45_exact"], default=None)
    ap.add_argument("--precision", type=int, choices=[32, 64], default=None)
    ap.add_argument("--schedule", choices=["direct", "queue"], default=None)
    ap.add_argument("--gpus", type=int, default=1, help="split the frame's rows over this many GPUs of the node")
    ap.add_argument("--full-trace", action="store_true",
                    help="trace every row instead of the reference's top-half trace + mirror (Kerr, equatorial observer)")
    args = ap.parse_args()
    main(M=args.M, a=args.a, r_obs_mult=args.r_obs, psi=(np.radians(args.psi_y), np.radians(args.psi_x)),
         vertical_fov_deg=args.fov_v, image_path=args.image, output_path=args.output, synthetic=args.synthetic,
         staged=args.staged, integrator=args.integrator, precision=args.precision, schedule=args.schedule,
         gpus=args.gpus, full_trace=args.full_trace, dedup_alpha=args.dedup_alpha, lookup_cache=args.lookup_cache)
