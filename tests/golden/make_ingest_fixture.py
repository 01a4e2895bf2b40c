#!/usr/bin/env python3
"""Writes tests/golden/ingest/reference_frames.npz: small driver messages (raw bytes) together with what the REFERENCE's own
Preprocess::process_cut_frame_pcl2 / process_cut_frame_livox / process (cut_frame_num 0) return for them.  The reference code is the unmodified
/root/reference/src/preprocess.cpp compiled by `make -C oracle ref` into oracle/_ref/libref_preprocess.so; this script
only runs where that library exists (this container).  Times are made distinct so that the reference's unstable std::sort
has a unique answer."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from harness import wire  # noqa: E402
from harness import synth
from oracle import oracle as O  # noqa: E402


def main():
    assert O.ref_preprocess_lib() is not None, "build oracle/_ref first: make -C oracle ref"
    hall = synth.Hall()
    xyz, ring, t_ms = wire.raw_sweep(hall, "tiny", synth.rot_zyx(0.01, 0.02, 0.4), np.array([0.5, -1.0, 0.2]))
    rng = np.random.default_rng(9)
    t_j = t_ms + rng.permutation(len(t_ms)) * 1e-3
    out = {}
    # (cut 0: Preprocess::process - the non-cutting handlers of initialization/cut_frame: false, src/preprocess.cpp:337-713)
    cases = [(wire.VELO, 3, 100, 1, True), (wire.OUSTER, 4, 100, 2, True), (wire.PANDAR, 2, 100, 1, True),
             (wire.ROBOSENSE, 3, 5, 1, True), (wire.VELO, 2, 100, 1, False),
             (wire.OUSTER, 0, 100, 2, True), (wire.VELO, 0, 100, 1, True), (wire.VELO, 0, 100, 2, False), (wire.L515, 0, 100, 3, True)]
    for k, (lt, cut, sc, pfn, with_time) in enumerate(cases):
        stamp, blind, n_scans = 321.25, 1.2, 14
        raw = wire.pack_pcl2(lt, xyz, ring, t_j, stamp, with_time=with_time)
        if not with_time and cut > 0:
            # the azimuth-derived times of different rings tie; keep one ring so that the reference's order is unique
            n_scans = 1
        fr = O.ref_ingest_pcl2(raw, len(xyz), wire.pc2_fields(lt), lt, n_scans, pfn, blind, stamp, cut, sc)
        c = f"case{k}"
        out[c + "/meta"] = np.array([0, lt, len(xyz), n_scans, pfn, cut, sc])
        out[c + "/params"] = np.array([blind, stamp])
        out[c + "/raw"] = np.frombuffer(raw, np.uint8)
        out[c + "/begin_ms"] = np.array([tb for tb, _ in fr])
        out[c + "/counts"] = np.array([len(p) for _, p in fr])
        out[c + "/points"] = np.concatenate([p for _, p in fr])
    raw, n = wire.avia_sweep(hall, synth.rot_zyx(0, 0, 0.3), np.array([0.5, 0.5, 0.0]), n_points=3000)
    for k, (cut, sc, pfn) in enumerate([(5, 100, 2), (3, 2, 1), (0, 100, 2)]):
        fr = O.ref_ingest_livox(raw, n, wire.livox_fields(), 6, pfn, 1.0, 12.5, cut, sc)
        c = f"livox{k}"
        out[c + "/meta"] = np.array([1, wire.AVIA, n, 6, pfn, cut, sc])
        out[c + "/params"] = np.array([1.0, 12.5])
        out[c + "/raw"] = np.frombuffer(raw, np.uint8)
        out[c + "/begin_ms"] = np.array([tb for tb, _ in fr])
        out[c + "/counts"] = np.array([len(p) for _, p in fr])
        out[c + "/points"] = np.concatenate([p for _, p in fr])
    path = os.path.join(HERE, "ingest", "reference_frames.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
