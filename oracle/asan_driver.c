/* ASan / UBSan driver for the CPU oracle (test infrastructure, `make -C oracle asan`): runs reset + control steps with
 * random actions, an explicit reset_idx and the stand-alone stage entry points on a config / model pair dumped by
 * tests/test_oracle_sanitizers.py, with the oracle compiled into the same sanitized executable (no LD_PRELOAD games). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/dexsim.h"

void* orc_create(const DexSimConfig* cfg, const DexHandModel* model);
void orc_destroy(void* h);
void orc_init_state(void* h);
void orc_reset(void* h);
void orc_step(void* h, const float* actions);
void orc_reset_idx(void* h, const int64_t* ids, int k);
void orc_physics_step(void* h);
void orc_substep(void* h, int last);
void orc_publish(void* h);
void orc_post_physics(void* h, int obs_only);
void orc_get_obs_buf(void* h, float* out);
void orc_get_stats(void* h, float* out);
int orc_get_contacts(void* h, int env, double* out);
int orc_set_field(void* h, const char* name, const double* in);
int orc_get_field(void* h, const char* name, double* out);
int orc_field_rows(const char* name);

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <structs.bin> <steps>\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("open"); return 2; }
  DexSimConfig cfg; DexHandModel model;
  if (fread(&cfg, sizeof cfg, 1, f) != 1 || fread(&model, sizeof model, 1, f) != 1) { fprintf(stderr, "short read\n"); return 2; }
  fclose(f);
  const int steps = atoi(argv[2]), n = cfg.num_envs, na = cfg.num_actions;
  void* h = orc_create(&cfg, &model);
  orc_init_state(h);
  orc_reset(h);
  float* act = (float*)malloc(sizeof(float) * (size_t)n * na);
  float* obs = (float*)malloc(sizeof(float) * (size_t)n * cfg.num_obs);
  uint32_t s = 12345u;
  double csum = 0;
  for (int t = 0; t < steps; t++) {
    for (int i = 0; i < n * na; i++) { s = s * 1664525u + 1013904223u; act[i] = 2.0f * (float)(s >> 8) / 16777216.0f - 1.0f; }
    orc_step(h, act);
    if (t == steps / 2) { int64_t ids[2] = {0, n - 1}; orc_reset_idx(h, ids, 2); }
    orc_get_obs_buf(h, obs);
    for (int i = 0; i < n * cfg.num_obs; i += 7) csum += obs[i];
  }
  /* hands lowered onto the box: the general contact path (narrowphase with hand contacts, dense rows, PGS) */
  {
    int rows = orc_field_rows("q");
    double* q = (double*)calloc((size_t)rows * n, sizeof(double));
    for (int e = 0; e < n; e++) { q[2 * n + e] = -0.40; for (int j = 6; j < rows; j++) q[j * n + e] = 0.05 * ((j + e) % 6); }
    orc_set_field(h, "q", q); orc_set_field(h, "targets", q);
    for (int t = 0; t < 4; t++) orc_physics_step(h);
    double c[DEXSIM_KMAX * 10];
    int k = orc_get_contacts(h, 0, c);
    csum += k;
    free(q);
  }
  orc_substep(h, 1); orc_publish(h); orc_post_physics(h, 1); orc_post_physics(h, 0);
  float st[DEXSIM_STAT_WORDS];
  orc_get_stats(h, st);
  printf("asan driver ok: %d envs, %d steps, checksum %.6f, mean contacts %.2f\n", n, steps, csum, st[DEXSIM_STAT_MEAN_CONTACTS]);
  free(act); free(obs);
  orc_destroy(h);
  return 0;
}
