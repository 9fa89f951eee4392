/*
 * oracle/voxel_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar, single-threaded CPU restatement of the LiDAR voxel path of
 * Advaith-Sajeev/LiDAR-Vision-VQA (vendored OpenPCDet 0.6.0), used ONLY as
 * the checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg.  The product path (the .hip files under lidar-vision-vqa_amd/csrc) never links,
 * imports or calls anything in this file.
 *
 * What it follows (reference file:line, relative to /root/reference):
 *   orc_voxelize_hard      src/lidar-encoder/pcdet/datasets/processor/data_processor.py:16-61,133-180
 *                          -> spconv.utils.Point2VoxelCPU3d.point_to_voxel / VoxelGeneratorV2.generate.
 *                          spconv is a THIRD-PARTY dependency that is neither vendored nor version
 *                          pinned by the reference (setup.py:48 comments it out, docs/INSTALL.md:9,30-33
 *                          allows v1.0 / v1.2 / v2.x).  The published algorithm is restated here
 *                          (spconv v2 `Point2VoxelCPU::point_to_voxel`, identical to the spconv v1.x
 *                          `points_to_voxel_3d_np` C++ kernel): per point, per axis j
 *                          c_j = floor((p_j - lo_j) / vs_j) in fp32, drop the point if c_j < 0 or
 *                          c_j >= grid_j, voxel id = first-appearance order through a dense
 *                          coor->voxel lookup table, keep the first T points per voxel in input
 *                          order, stop CREATING voxels at max_voxels (`continue`, default) or stop
 *                          the whole scan there (`break`, spconv-1.0 / second.pytorch numba lineage).
 *                          PARITY UNPINNED for this one function: the reference holds no test or
 *                          golden vector at this boundary and spconv cannot run here; it is pinned by
 *                          hand-checkable known-answer tests and by the hard-vs-dynamic voxel-set
 *                          identity (tests/test_oracle_voxel.py).
 *   orc_dynamic_keys       src/lidar-encoder/pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:53-59
 *                          (== dynamic_voxel_vfe.py:60-69; 2-D form dynamic_pillar_vfe.py:93-103)
 *   orc_mean_vfe           src/lidar-encoder/pcdet/models/backbones_3d/vfe/mean_vfe.py:25-29
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off (see oracle/Makefile).  -ffp-contract=off and no
 * -ffast-math so that (p - lo) / vs is an IEEE fp32 subtract followed by an IEEE fp32 divide.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* Hard voxeliser, one scene.
 *  pts        [n, c] fp32, row-major; columns 0..2 are x,y,z
 *  range      [6] fp32  (lo_x, lo_y, lo_z, hi_x, hi_y, hi_z)
 *  vsize      [3] fp32
 *  grid       [3] int32 (nx, ny, nz)
 *  lut        [nz*ny*nx] int32, must be all -1 on entry; restored to all -1 on exit
 *             (spconv keeps the same persistent dense table and resets only touched cells)
 *  voxels     [max_voxels, max_pts, c] fp32, zero-filled here for the voxels that are created
 *  coords     [max_voxels, 3] int32 (z, y, x)
 *  num_pts    [max_voxels] int32
 *  break_on_cap 0: `continue` at the cap (spconv >= 1.1 / 2.x, default); 1: `break`
 * returns the number of voxels. */
int orc_voxelize_hard(const float *pts, int64_t n, int c, const float *range, const float *vsize,
                      const int32_t *grid, int max_pts, int max_voxels, int break_on_cap,
                      int32_t *lut, float *voxels, int32_t *coords, int32_t *num_pts)
{
    const int nx = grid[0], ny = grid[1];
    int voxel_num = 0;
    for (int64_t i = 0; i < n; ++i) {
        const float *p = pts + i * c;
        int cc[3];
        int failed = 0;
        for (int j = 0; j < 3; ++j) {
            volatile float d = p[j] - range[j];  /* volatile: keep the fp32 rounding of the subtract */
            float q = d / vsize[j];
            int ci = (int)floorf(q);
            if (ci < 0 || ci >= grid[j]) { failed = 1; break; }
            cc[j] = ci;
        }
        if (failed) continue;
        const int64_t cell = ((int64_t)cc[2] * ny + cc[1]) * nx + cc[0];
        int vid = lut[cell];
        if (vid == -1) {
            if (voxel_num >= max_voxels) {
                if (break_on_cap) break;
                continue;
            }
            vid = voxel_num++;
            lut[cell] = vid;
            coords[vid * 3 + 0] = cc[2];
            coords[vid * 3 + 1] = cc[1];
            coords[vid * 3 + 2] = cc[0];
            num_pts[vid] = 0;
            memset(voxels + (int64_t)vid * max_pts * c, 0, sizeof(float) * (size_t)max_pts * c);
        }
        int k = num_pts[vid];
        if (k < max_pts) {
            memcpy(voxels + ((int64_t)vid * max_pts + k) * c, p, sizeof(float) * (size_t)c);
            num_pts[vid] = k + 1;
        }
    }
    for (int v = 0; v < voxel_num; ++v) {
        const int64_t cell = ((int64_t)coords[v * 3] * ny + coords[v * 3 + 1]) * nx + coords[v * 3 + 2];
        lut[cell] = -1;
    }
    return voxel_num;
}

/* Dynamic voxelisation keys (the part of the reference that is pure arithmetic).
 *  pts   [n, c] fp32 with column 0 = batch index, 1..3 = x,y,z   (pcdet `points` after collate_batch)
 *  ndim  3: key = b*S_xyz + cx*S_yz + cy*S_z + cz  (dynamic_mean_vfe.py:56-59)
 *        2: key = b*S_xy  + cx*S_y  + cy           (dynamic_pillar_vfe.py:99-101; z is not range-tested)
 *  The reference does this arithmetic in int32 (torch .int() tensors times python ints), so it wraps;
 *  it is restated with uint32 wrap-around and returned as int32.
 *  keys  [n] int32 out (undefined where valid==0);  cxyz [n,3] int32 out;  valid [n] uint8 out
 * returns the number of valid points. */
int64_t orc_dynamic_keys(const float *pts, int64_t n, int c, const float *range, const float *vsize,
                         const int32_t *grid, int ndim, int32_t *keys, int32_t *cxyz, uint8_t *valid)
{
    int64_t nvalid = 0;
    const uint32_t gx = (uint32_t)grid[0], gy = (uint32_t)grid[1], gz = (uint32_t)grid[2];
    for (int64_t i = 0; i < n; ++i) {
        const float *p = pts + i * c;
        int cc[3] = {0, 0, 0};
        int ok = 1;
        for (int j = 0; j < ndim; ++j) {
            volatile float d = p[1 + j] - range[j];
            float q = d / vsize[j];
            /* torch: floor() then .int(); values here are far inside int32 for any sane cloud */
            float f = floorf(q);
            int ci = (f >= 2147483648.0f || f < -2147483648.0f || f != f) ? INT32_MIN : (int)f;
            cc[j] = ci;
            if (ci < 0 || ci >= grid[j]) ok = 0;
        }
        cxyz[i * 3 + 0] = cc[0]; cxyz[i * 3 + 1] = cc[1]; cxyz[i * 3 + 2] = cc[2];
        valid[i] = (uint8_t)ok;
        if (!ok) continue;
        ++nvalid;
        const uint32_t b = (uint32_t)(int32_t)p[0];
        uint32_t k;
        if (ndim == 3) k = b * (gx * gy * gz) + (uint32_t)cc[0] * (gy * gz) + (uint32_t)cc[1] * gz + (uint32_t)cc[2];
        else           k = b * (gx * gy) + (uint32_t)cc[0] * gy + (uint32_t)cc[1];
        keys[i] = (int32_t)k;
    }
    return nvalid;
}

/* MeanVFE: sum over ALL T slots (padding is zero) divided by clamp_min(count, 1). */
void orc_mean_vfe(const float *voxels, const int32_t *num_pts, int64_t m, int t, int c, float *out)
{
    for (int64_t v = 0; v < m; ++v) {
        float norm = (float)(num_pts[v] < 1 ? 1 : num_pts[v]);
        for (int k = 0; k < c; ++k) {
            float s = 0.f;
            for (int j = 0; j < t; ++j) s += voxels[(v * t + j) * c + k];
            out[v * c + k] = s / norm;
        }
    }
}
