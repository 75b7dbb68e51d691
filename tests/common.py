"""Shared helpers for the parity tests: run the same frame through libfovpt (GPU) and the oracle."""
import numpy as np

from fovpathtracing_optixcodelatest_amd import abi, renderer, scenes


def make_gpu(model, probe_data, camera, size, cfg, gaze=None, subframe_index=0):
    r = renderer.SampleRenderer(model)
    r.resize(size)
    r.setCamera(renderer.Camera(camera["eye"], camera["lookat"], camera["up"], camera["fovy"], size[0] / float(size[1])))
    r.setProbe(renderer.ProbeData(probe_data).BuildCDF())
    r.config = cfg
    gx, gy = gaze if gaze is not None else (size[0] // 2, size[1] // 2)
    r.launchParams.frame.c.x, r.launchParams.frame.c.y = gx, gy
    r.launchParams.frame.subframe_index = subframe_index
    return r


def make_oracle(orc, model, probe_data, camera, size, gaze=None, subframe_index=0):
    S = orc.OracleScene(model)
    probe = orc.HostProbe(probe_data)
    F = orc.OracleFrame(size[0], size[1], probe, camera, gaze=gaze, subframe_index=subframe_index)
    return S, F


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = np.sqrt((b * b).sum())
    return float(np.sqrt(((a - b) ** 2).sum()) / den) if den > 0 else float(np.sqrt(((a - b) ** 2).sum()))


def compare_frames(gpu_accum, gpu_frame, ora_accum, ora_frame):
    """Returns (rel_l2 of rgb radiance, #pixels whose float4 differs in any bit, #rgba8 mismatches)."""
    l2 = rel_l2(gpu_accum[..., :3], ora_accum[..., :3])
    bits = int((gpu_accum.view(np.uint32) != ora_accum.view(np.uint32)).any(axis=-1).sum())
    px = int((gpu_frame != ora_frame).sum())
    return l2, bits, px


def cfg_foveated(r_inner, r_outer, spp=(1, 2, 8), max_depth=4):
    c = abi.Config.reference_default()
    c.r_inner, c.r_outer = r_inner, r_outer
    c.spp_periphery, c.spp_middle, c.spp_fovea = spp
    c.max_depth = max_depth
    return c


def cfg_uniform(spp=4, max_depth=4):
    c = abi.Config.reference_default()
    c.uniform = 1
    c.spp_uniform = spp
    c.max_depth = max_depth
    return c


def encode_hdr_rle(texels_rgbe, comments=("# written by tests/common.py",), magic=b"#?RADIANCE"):
    """A Radiance .hdr file (new-style RLE scanlines) holding the given (H, W, 4) uint8 RGBE texels."""
    h, w, _ = texels_rgbe.shape
    assert 8 <= w < 32768
    out = bytearray(magic + b"\n")
    for c in comments:
        out += c.encode() + b"\n"
    out += b"FORMAT=32-bit_rle_rgbe\n\n" + ("-Y %d +X %d\n" % (h, w)).encode()
    for j in range(h):
        out += bytes([2, 2, w >> 8, w & 255])
        for k in range(4):
            row = texels_rgbe[j, :, k]
            i = 0
            while i < w:
                run = 1
                while i + run < w and run < 127 and row[i + run] == row[i]:
                    run += 1
                if run >= 3:
                    out += bytes([128 + run, int(row[i])])
                    i += run
                else:
                    n = 1
                    while i + n < w and n < 128 and not (i + n + 2 < w and row[i + n] == row[i + n + 1] == row[i + n + 2]):
                        n += 1
                    out += bytes([n]) + bytes(row[i:i + n].tolist())
                    i += n
    return bytes(out)


def encode_png(samples: np.ndarray, depth: int, ctype: int, filters=(0, 1, 2, 3, 4), interlace: bool = False,
               plte: bytes = None, trns: bytes = None) -> bytes:
    """A small PNG writer for loader tests: samples (H, W, C) integers of `depth` bits, every scanline
    filtered with the next entry of `filters` (so each filter type is exercised), optional Adam7."""
    import struct
    import zlib
    h, w, chans = samples.shape
    bpp = max(1, chans * depth // 8)

    def pack(block: np.ndarray) -> list:                         # (ph, pw, C) -> list of byte rows
        rows = []
        for r in block:
            flat = r.reshape(-1).astype(np.uint32)
            if depth == 8:
                rows.append(flat.astype(np.uint8).tolist())
            elif depth == 16:
                rows.append(np.stack([flat >> 8, flat & 255], axis=1).reshape(-1).astype(np.uint8).tolist())
            else:
                bits = ((flat[:, None] >> np.arange(depth - 1, -1, -1)) & 1).reshape(-1).astype(np.uint8)
                rows.append(np.packbits(bits).tolist())
        return rows

    counter = [0]

    def filtered(rows: list) -> bytes:
        out = bytearray()
        prev = [0] * (len(rows[0]) if rows else 0)
        for cur in rows:
            ft = filters[counter[0] % len(filters)]
            counter[0] += 1
            out.append(ft)
            for i, x in enumerate(cur):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ft == 0:
                    p = 0
                elif ft == 1:
                    p = a
                elif ft == 2:
                    p = b
                elif ft == 3:
                    p = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                out.append((x - p) & 255)
            prev = cur
        return bytes(out)

    if interlace:
        body = b""
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                body += filtered(pack(sub))
    else:
        body = filtered(pack(samples))

    def chunk(kind: bytes, payload: bytes) -> bytes:
        return struct.pack(">I", len(payload)) + kind + payload + struct.pack(">I", zlib.crc32(kind + payload))

    z = zlib.compress(body)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    out += chunk(b"gAMA", struct.pack(">I", 45455))              # ignored by stb_image
    if plte is not None:
        out += chunk(b"PLTE", plte)
    if trns is not None:
        out += chunk(b"tRNS", trns)
    half = len(z) // 2
    return out + chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")


# ---- Truevision TGA writer for the loader tests and tests/golden/make_ref_golden.py ----
def encode_tga(w, h, itype, bits, px_bytes, desc=0, cmap=b"", cm_len=0, cm_bits=0, idfield=b"id"):
    import struct
    return struct.pack("<BBBHHBHHHHBB", len(idfield), 1 if cmap else 0, itype, 0, cm_len, cm_bits, 0, 0, w, h, bits, desc) + idfield + cmap + px_bytes


def tga_rle(px: np.ndarray) -> bytes:                                # (n, nb) -> alternating run / literal packets
    out, i, n = bytearray(), 0, len(px)
    toggle = True
    while i < n:
        cnt = min(n - i, 3 if toggle else 5)
        if toggle:
            px[i:i + cnt] = px[i]
            out += bytes([128 | (cnt - 1)]) + px[i].tobytes()
        else:
            out += bytes([cnt - 1]) + px[i:i + cnt].tobytes()
        i += cnt
        toggle = not toggle
    return bytes(out)
