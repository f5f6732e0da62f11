/* Writes the MVP-layout HDF5 fixtures of tests/test_h5lite.py with the REAL HDF5 library
 * (the image carries libhdf5 1.10 under /opt/conda; h5py itself is absent).  Layout
 * follows completion/README.md:21-32 / completion/dataset.py:21-34 of the reference:
 *   incomplete_pcds (26*S, P, 3) f32, complete_pcds (S, P, 3) f32, labels (26*S,) i64
 * with S = 2 shapes and P = 8 points.  Values are a closed formula so the test can
 * recompute them:  incomplete[i][j][k] = i + (3 j + k) / 64,  complete[s][j][k] = -(s + (3 j + k) / 64),
 * labels[i] = (i / 26) * 5 + 3.
 * Three files: default property lists (what h5py's create_dataset(data=) writes: superblock v0,
 * contiguous), chunked + shuffle + gzip (h5py compression="gzip", shuffle=True), and
 * libver=latest (superblock v3 / object header v2 / link messages).
 * Build + run: tests/golden/make_h5_fixtures.sh */
#include <hdf5.h>
#include <stdint.h>
#include <stdio.h>

#define S 2
#define V 26
#define P 8

static float inc[S * V][P][3], com[S][P][3];
static int64_t lab[S * V];

static void write_file(const char *path, int mode) {
  hid_t fapl = H5Pcreate(H5P_FILE_ACCESS);
  if (mode == 2) H5Pset_libver_bounds(fapl, H5F_LIBVER_LATEST, H5F_LIBVER_LATEST);
  hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, fapl);
  hsize_t d3[3] = {S * V, P, 3}, c3[3] = {S, P, 3}, d1[1] = {S * V};
  hid_t dcpl3 = H5Pcreate(H5P_DATASET_CREATE), dcpl1 = H5Pcreate(H5P_DATASET_CREATE);
  if (mode == 1) {
    hsize_t ch3[3] = {7, 5, 3}, ch1[1] = {20};   /* chunks that do not divide the extents */
    H5Pset_chunk(dcpl3, 3, ch3); H5Pset_shuffle(dcpl3); H5Pset_deflate(dcpl3, 4);
    H5Pset_chunk(dcpl1, 1, ch1); H5Pset_shuffle(dcpl1); H5Pset_deflate(dcpl1, 4);
  }
  hid_t sp = H5Screate_simple(3, d3, NULL);
  hid_t ds = H5Dcreate2(f, "incomplete_pcds", H5T_IEEE_F32LE, sp, H5P_DEFAULT, dcpl3, H5P_DEFAULT);
  H5Dwrite(ds, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, inc); H5Dclose(ds); H5Sclose(sp);
  sp = H5Screate_simple(3, c3, NULL);
  hid_t dc = mode == 1 ? H5Pcreate(H5P_DATASET_CREATE) : dcpl3;
  if (mode == 1) { hsize_t ch[3] = {1, 8, 3}; H5Pset_chunk(dc, 3, ch); H5Pset_deflate(dc, 9); }
  ds = H5Dcreate2(f, "complete_pcds", H5T_IEEE_F32LE, sp, H5P_DEFAULT, dc, H5P_DEFAULT);
  H5Dwrite(ds, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, com); H5Dclose(ds); H5Sclose(sp);
  sp = H5Screate_simple(1, d1, NULL);
  ds = H5Dcreate2(f, "labels", H5T_STD_I64LE, sp, H5P_DEFAULT, dcpl1, H5P_DEFAULT);
  H5Dwrite(ds, H5T_NATIVE_INT64, H5S_ALL, H5S_ALL, H5P_DEFAULT, lab); H5Dclose(ds); H5Sclose(sp);
  H5Fclose(f);
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s <outdir>\n", argv[0]); return 2; }
  for (int i = 0; i < S * V; ++i) {
    lab[i] = (i / V) * 5 + 3;
    for (int j = 0; j < P; ++j) for (int k = 0; k < 3; ++k) inc[i][j][k] = (float)i + (float)(3 * j + k) / 64.0f;
  }
  for (int s = 0; s < S; ++s)
    for (int j = 0; j < P; ++j) for (int k = 0; k < 3; ++k) com[s][j][k] = -((float)s + (float)(3 * j + k) / 64.0f);
  char path[512];
  const char *names[3] = {"mvp_tiny_default.h5", "mvp_tiny_gzip.h5", "mvp_tiny_latest.h5"};
  for (int m = 0; m < 3; ++m) { snprintf(path, sizeof path, "%s/%s", argv[1], names[m]); write_file(path, m); }
  return 0;
}
