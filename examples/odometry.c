/*
 * examples/odometry.c -- the smallest host program on top of the C-ABI (include/suma_hip.h):
 * SuMa++ odometry over a directory of KITTI velodyne scans, one pose per line on stdout
 * (KITTI devkit format: the upper 3x4 of the sensor pose, row-major).
 *
 *   cc -O2 -Iinclude examples/odometry.c -Lsemantic_suma_amd -lsuma_hip -Wl,-rpath,$PWD/semantic_suma_amd -o odometry
 *   ./odometry /data/kitti/sequences/00/velodyne 4541 [labels_dir]
 *
 * Scan files: <dir>/%06d.bin, N x 4 float32 (x, y, z, remission) as read by the reference's
 * KITTIReader (src/io/KITTIReader.cpp:140-167).  Optional SemanticKITTI labels: <labels_dir>/%06d.label,
 * N x uint32 (lower 16 bits = class id); without them the run is plain SuMa (labels 0, probabilities 0).
 * This mirrors what SurfelMapping::processScan is fed by the reference's visualizer (src/visualizer/visualizer.cpp)
 * minus RangeNet++ inference, which is outside the hot path this library replaces.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "suma_hip.h"

static size_t file_size(FILE* f) {
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  return n < 0 ? 0 : (size_t)n;
}

int main(int argc, char** argv) {
  if (argc < 3) {
    fprintf(stderr, "usage: %s <velodyne_dir> <n_scans> [labels_dir]\n", argv[0]);
    return 2;
  }
  const char* dir = argv[1];
  const int n_scans = atoi(argv[2]);
  const char* label_dir = argc > 3 ? argv[3] : NULL;

  suma_params p;
  suma_params_default(&p); /* config/default.xml of the reference; 64 x 900 images */
  p.data_width = p.model_width = 2048;

  suma_pipeline* pipe = NULL;
  if (suma_pipeline_create(&p, /*hip_device=*/0, &pipe) != SUMA_OK) {
    fprintf(stderr, "suma_pipeline_create: %s\n", suma_last_error(NULL));
    return 1;
  }

  suma_float4* pts = NULL;
  float *labels = NULL, *probs = NULL;
  uint32_t* raw = NULL;
  size_t cap = 0;
  char path[4096];
  for (int k = 0; k < n_scans; ++k) {
    snprintf(path, sizeof(path), "%s/%06d.bin", dir, k);
    FILE* f = fopen(path, "rb");
    if (!f) {
      fprintf(stderr, "cannot open %s\n", path);
      break;
    }
    const size_t n = file_size(f) / (4 * sizeof(float));
    if (n > cap) {
      cap = n;
      pts = (suma_float4*)realloc(pts, cap * sizeof(suma_float4));
      labels = (float*)realloc(labels, cap * sizeof(float));
      probs = (float*)realloc(probs, cap * sizeof(float));
      raw = (uint32_t*)realloc(raw, cap * sizeof(uint32_t));
    }
    if (fread(pts, sizeof(suma_float4), n, f) != n) n == 0 ? (void)0 : (void)fprintf(stderr, "short read: %s\n", path);
    fclose(f);
    memset(labels, 0, n * sizeof(float));
    memset(probs, 0, n * sizeof(float));
    if (label_dir) {
      snprintf(path, sizeof(path), "%s/%06d.label", label_dir, k);
      FILE* g = fopen(path, "rb");
      if (g) {
        const size_t m = fread(raw, sizeof(uint32_t), n, g);
        fclose(g);
        for (size_t i = 0; i < m; ++i) {
          labels[i] = (float)(raw[i] & 0xffffu);
          probs[i] = 1.0f; /* ground-truth labels: full confidence */
        }
      }
    }
    /* fixed_iterations = 0: the stopping tests of LieGaussNewton decide (max iterations from the params) */
    const int r = suma_pipeline_process_scan(pipe, pts, labels, probs, (uint32_t)n, 0);
    if (r != SUMA_OK) {
      fprintf(stderr, "scan %d: %s\n", k, suma_last_error(suma_pipeline_ctx(pipe)));
      break;
    }
    double T[16]; /* column-major */
    suma_pipeline_pose(pipe, T);
    printf("%.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", T[0], T[4], T[8], T[12], T[1], T[5], T[9],
           T[13], T[2], T[6], T[10], T[14]);
  }
  fprintf(stderr, "frame-to-frame fallbacks: %u\n", suma_pipeline_track_loss(pipe));
  free(pts);
  free(labels);
  free(probs);
  free(raw);
  suma_pipeline_destroy(pipe);
  return 0;
}
