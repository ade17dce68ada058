/*
 * knn_oracle.c -- CPU restatement of simple-knn's distCUDA2 contract.
 * TEST INFRASTRUCTURE ONLY (see raster_oracle.c header).
 *
 * Contract (KNN/ = /root/reference/submodules/simple-knn/): for every point, the mean of the
 * 3 smallest squared Euclidean distances to OTHER points, written at the point's original
 * index (KNN/simple_knn.cu:147-183, :182; updateKBest<3> :131-145).  The reference's Morton
 * ordering + box pruning is an acceleration structure only; the result is the exact 3-NN mean,
 * so the oracle is the brute-force definition.  Distances are float32, d = dx*dx+dy*dy+dz*dz
 * (KNN/simple_knn.cu:119-129).  When fewer than 3 other points exist the unfilled slots keep
 * FLT_MAX (KNN/simple_knn.cu:160: best initialised to FLT_MAX).
 */
#include <float.h>
#include <stdint.h>

void oracle_dist2(int P, const float* pts, float* out)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        float best[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
        float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            float dx = pts[3 * j] - px, dy = pts[3 * j + 1] - py, dz = pts[3 * j + 2] - pz;
            float d = dx * dx + dy * dy + dz * dz;
            /* updateKBest<3>: insertion keeping ascending order */
            for (int k = 0; k < 3; k++) {
                if (best[k] > d) { float t = best[k]; best[k] = d; d = t; }
            }
        }
        out[i] = (best[0] + best[1] + best[2]) / 3.0f;
    }
}
