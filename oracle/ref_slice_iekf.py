#!/usr/bin/env python3
"""ORACLE build step - TEST INFRASTRUCTURE ONLY.

Cuts the per-scan update out of the reference's src/laserMapping.cpp AT BUILD TIME, so that the reference's own text - the
residual / selection loop, the compaction, the Jacobian rows, the literal 24 x m gain, the convergence / re-match schedule, the
covariance update (src/laserMapping.cpp:936-1134, inline in main()) and map_incremental (:516-559) - compiles into
oracle/_ref/libref_iekf.so around oracle/ref_iekf_wrap.cpp and pins oracle/orc_iekf.hpp (tests/test_oracle_iekf_pinned.py).

Nothing of the reference enters the repository: the slices are written to oracle/_ref/gen/ (git-ignored, like every other output
of `make ref`), and they are cut by ANCHOR TEXT - a statement that opens the piece, brace counting to its end - not by line
numbers; the line ranges found are printed and checked against the ranges SURVEY.md section 8 cites, so a reference that moved
fails the build instead of slicing something else.

usage: ref_slice_iekf.py <reference root> <output directory>
"""
import os
import re
import sys


def block_from(lines, start, what):
    """lines[start] opens a brace block (the first '{' at or after it): returns the index of the line that closes it."""
    depth, seen = 0, False
    for k in range(start, len(lines)):
        code = re.sub(r"//.*", "", lines[k])
        code = re.sub(r'"(\\.|[^"\\])*"', '""', code)
        for ch in code:
            if ch == "{":
                depth += 1
                seen = True
            elif ch == "}":
                depth -= 1
                if seen and depth == 0:
                    return k
    raise SystemExit(f"ref_slice_iekf: no end of block for {what}")


def find(lines, text, what, after=0):
    for k in range(after, len(lines)):
        if text in lines[k]:
            return k
    raise SystemExit(f"ref_slice_iekf: anchor not found for {what}: {text!r}")


def main():
    ref, out = sys.argv[1], sys.argv[2]
    src = os.path.join(ref, "src", "laserMapping.cpp")
    lines = open(src).read().split("\n")
    os.makedirs(out, exist_ok=True)
    pieces = {}

    def function(name, anchor, expect):
        a = find(lines, anchor, name)
        b = block_from(lines, a, name)
        pieces[name] = (a, b, expect)

    # file-scope functions, whole
    function("calc_dist", "float calc_dist(PointType p1, PointType p2) {", (153, 156))
    function("calc_body_var", "void calcBodyVar(Eigen::Vector3d &pb, const float range_inc,", (158, 181))
    function("point_body_to_world", "void pointBodyToWorld(PointType const *const pi, PointType *const po) {", (209, 220))
    function("map_incremental", "void map_incremental() {", (516, 559))
    # main(): the declarations in front of the loop, their initialisation, and the update itself
    a = find(lines, "/*** variables definition ***/", "declarations")
    b = find(lines, "bool flg_EKF_converged, EKF_stop_flg = 0;", "declarations", a)
    pieces["main_decls"] = (a + 1, b, (808, 816))
    a = find(lines, "G.setZero();", "initialisation")
    pieces["main_init"] = (a, a + 2, (841, 843))
    assert "I_STATE.setIdentity();" in lines[a + 2], "initialisation block moved"
    a = find(lines, "/*** ICP and iterated Kalman filter update ***/", "update")
    f = find(lines, "for (iterCount = 0; iterCount < NUM_MAX_ITERATIONS; iterCount++) {", "update loop", a)
    b = block_from(lines, f, "update loop")
    pieces["main_update"] = (a, b, (936, 1134))
    # file-scope variables the pieces use, each by the line that declares it (verbatim)
    wanted = ["int iterCount = 0, feats_down_size = 0, NUM_MAX_ITERATIONS = 0", "double res_mean_last = 0.05;",
              "double filter_size_surf_min = 0, filter_size_map_min = 0;", "double cube_len = 0, total_distance = 0,",
              "int kdtree_size_st = 0, kdtree_size_end = 0, add_point_size = 0;",
              "bool lidar_pushed, flg_reset, flg_exit = false, flg_EKF_inited = true;", "bool imu_en = false;",
              "vector<vector<int>> pointSearchInd_surf;", "vector<PointVector> Nearest_Points;",
              "bool point_selected_surf[100000] = {0};", "float res_last[100000] = {0.0};", "double total_residual;",
              "PointCloudXYZI::Ptr feats_down_body(new PointCloudXYZI());", "PointCloudXYZI::Ptr feats_down_world(new PointCloudXYZI());",
              "PointCloudXYZI::Ptr normvec(new PointCloudXYZI(100000, 1));", "PointCloudXYZI::Ptr laserCloudOri(new PointCloudXYZI(100000, 1));",
              "PointCloudXYZI::Ptr corr_normvect(new PointCloudXYZI(100000, 1));", "KD_TREE ikdtree;", "V3D euler_cur;",
              "V3D position_last(Zero3d);", "StatesGroup state;", "geometry_msgs::Quaternion geoQuat;"]
    glob = []
    first_fn = find(lines, "float calc_dist(PointType p1, PointType p2) {", "end of the file-scope variables")
    for w in wanted:
        k = find(lines, w, "file-scope variable")
        assert k < first_fn, w
        text = lines[k]
        while text.rstrip().endswith("\\"):  # (the `int iterCount ...` declaration continues on the next line)
            k += 1
            text = text.rstrip()[:-1] + lines[k]
        glob.append(text)
    with open(os.path.join(out, "globals.inc"), "w") as fh:
        fh.write("\n".join(glob) + "\n")
    report = []
    for name, (a, b, expect) in pieces.items():
        got = (a + 1, b + 1)
        report.append(f"{name}: src/laserMapping.cpp:{got[0]}-{got[1]}")
        if got != expect:
            raise SystemExit(f"ref_slice_iekf: {name} found at {got}, expected {expect}: the reference is not the surveyed one")
        with open(os.path.join(out, name + ".inc"), "w") as fh:
            fh.write("\n".join(lines[a:b + 1]) + "\n")
    # include/common_lib.h is compiled as it lies; nothing to cut there
    print("ref_slice_iekf: " + "; ".join(report))


if __name__ == "__main__":
    main()
