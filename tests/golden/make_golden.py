#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.

The reference has no tests or golden vectors for this path and cannot be built here, so the
fixtures come from two sources, both reproducible with this script:

  rng_kat.json     integer known answers for tea<4>, lcg and class Random, computed by an
                   INDEPENDENT pure-Python restatement (arbitrary-precision ints masked to 32 bit)
                   of cuda/random.h:34-59 and PT_sv5_/maths.h:170-227 -- not by the C++ oracle.
  frames.npz       regression images rendered by the C++ oracle (detmath mode) after it passed the
                   KATs and its BVH-vs-brute-force self checks: Cornell 64x64 uniform 4 spp depth 3,
                   and a 128x72 three-pass foveated frame.  They pin the oracle against drift and give
                   the GPU tests a committed target.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
M = 0xFFFFFFFF


def tea4(v0, v1):
    s0 = 0
    for _ in range(4):
        s0 = (s0 + 0x9E3779B9) & M
        v0 = (v0 + ((((v1 << 4) & M) + 0xA341316C) & M ^ ((v1 + s0) & M) ^ (((v1 >> 5) + 0xC8013EA4) & M))) & M
        v1 = (v1 + ((((v0 << 4) & M) + 0xAD90777D) & M ^ ((v0 + s0) & M) ^ (((v0 >> 5) + 0x7E95761E) & M))) & M
    return v0


def lcg_stream(seed, n):
    out = []
    for _ in range(n):
        seed = (1664525 * seed + 1013904223) & M
        out.append(seed & 0x00FFFFFF)
    return out


def random_stream(seed, n):
    s1 = (315645664 + seed) & M
    s2 = s1 ^ 0x13AB45FE
    out = []
    for _ in range(n):
        rot5 = ((s1 << 5) | (s1 >> 27)) & M
        s1 = ((s2 ^ rot5) ^ ((s1 * s2) & M)) & M
        rot12 = ((s2 << 12) | (s2 >> 20)) & M
        s2 = (s1 ^ rot12) & M
        out.append(s1)
    return out


def main():
    seeds = [0, 1, 2, 12345, 0x7FFFFFFF, 0x80000000, 0xDEADBEEF, 0xFFFFFFFF]
    kat = {
        "tea4": [[a, b, tea4(a, b)] for a in seeds for b in (0, 1, 7, 0xFFFFFFFF)],
        "lcg": {str(s): lcg_stream(s, 16) for s in seeds},
        "random": {str(s): random_stream(s, 16) for s in seeds},
    }
    with open(os.path.join(HERE, "rng_kat.json"), "w") as f:
        json.dump(kat, f, indent=0)

    from fovpathtracing_optixcodelatest_amd import abi, scenes
    from oracle import oracle_py as orc
    orc.build()
    orc.set_math_mode(True)
    out = {}
    S = orc.OracleScene(scenes.cornell_box())
    F = orc.OracleFrame(64, 64, orc.HostProbe(scenes.ambient_probe(64, 32, 0.2)), scenes.CORNELL_CAMERA)
    cfg = abi.Config.reference_default()
    cfg.uniform, cfg.spp_uniform, cfg.max_depth = 1, 4, 3
    cnt = orc.render(S, F, cfg, nthreads=1)
    out["cornell_accum"], out["cornell_frame"], out["cornell_counts"] = F.accum.copy(), F.frame.copy(), np.array(cnt, np.uint64)
    F = orc.OracleFrame(128, 72, orc.HostProbe(scenes.sky_probe()), scenes.CORNELL_CAMERA)
    cfg = abi.Config.reference_default()
    cfg.r_inner, cfg.r_outer = 10, 30
    cfg.spp_periphery, cfg.spp_middle, cfg.spp_fovea = 1, 2, 8
    cnt = orc.render(S, F, cfg, nthreads=1)
    out["fov_accum"], out["fov_frame"], out["fov_counts"] = F.accum.copy(), F.frame.copy(), np.array(cnt, np.uint64)
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **out)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
