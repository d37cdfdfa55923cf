/* Test-only: writes rank-3 datasets with the REAL libhdf5 (1.10.x from /opt/conda) in the layouts the built-in reader
 * must understand.  usage: h5_libhdf5_writer out.h5 variant nz ny nx [dtype: u8|u16|f32]
 * The voxel values are a fixed function of (z, y, x) that the test recomputes. */
#include <hdf5.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double value(hsize_t z, hsize_t y, hsize_t x) { return (double)((z * 7 + y * 3 + x * 5 + (z * y) % 11) % 251); }

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const char* path = argv[1];
  const char* var = argv[2];
  hsize_t dims[3] = {(hsize_t)atoi(argv[3]), (hsize_t)atoi(argv[4]), (hsize_t)atoi(argv[5])};
  const char* dt = argc > 6 ? argv[6] : "u8";
  const int latest = strncmp(var, "old_", 4) != 0;
  hid_t fapl = H5Pcreate(H5P_FILE_ACCESS);
  if (latest) H5Pset_libver_bounds(fapl, H5F_LIBVER_LATEST, H5F_LIBVER_LATEST);
  hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, fapl);
  hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
  hsize_t maxd[3] = {dims[0], dims[1], dims[2]};
  hsize_t chunk[3] = {4, 10, 14};
  const char* name = "image";
  hid_t loc = f;
  if (strstr(var, "contig")) {
  } else if (strstr(var, "compact")) {
    H5Pset_layout(dcpl, H5D_COMPACT);
  } else if (strstr(var, "single")) {
    chunk[0] = dims[0]; chunk[1] = dims[1]; chunk[2] = dims[2];
    H5Pset_chunk(dcpl, 3, chunk);
  } else if (strstr(var, "implicit")) {
    H5Pset_chunk(dcpl, 3, chunk);
    H5Pset_alloc_time(dcpl, H5D_ALLOC_TIME_EARLY);
  } else if (strstr(var, "paged")) {
    chunk[0] = 2; chunk[1] = 4; chunk[2] = 4;
    H5Pset_chunk(dcpl, 3, chunk);
  } else if (strstr(var, "unlimz")) {
    maxd[0] = H5S_UNLIMITED;
    H5Pset_chunk(dcpl, 3, chunk);
  } else if (strstr(var, "unlimx")) {
    maxd[2] = H5S_UNLIMITED;
    H5Pset_chunk(dcpl, 3, chunk);
  } else if (strstr(var, "unlim2")) {
    maxd[0] = H5S_UNLIMITED; maxd[1] = H5S_UNLIMITED;
    H5Pset_chunk(dcpl, 3, chunk);
  } else if (strstr(var, "maxbig")) {
    maxd[0] = dims[0] * 3; maxd[1] = dims[1] + 7;
    H5Pset_chunk(dcpl, 3, chunk);
  } else {  // "chunk": fixed dimensions -> fixed-array index (libver latest) or v1 B-tree (earliest)
    H5Pset_chunk(dcpl, 3, chunk);
  }
  if (strstr(var, "tiny")) { chunk[0] = 1; chunk[1] = 4; chunk[2] = 5; H5Pset_chunk(dcpl, 3, chunk); }  /* many chunks: deep index structures */
  if (strstr(var, "shuf")) H5Pset_shuffle(dcpl);
  if (strstr(var, "gz")) H5Pset_deflate(dcpl, 4);
  if (strstr(var, "unlim2tiny")) { chunk[0] = 1; chunk[1] = 2; chunk[2] = 5; H5Pset_chunk(dcpl, 3, chunk); }  /* thousands of B-tree records: internal nodes */
  if (strstr(var, "nested")) {
    hid_t g1 = H5Gcreate2(f, "g1", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    loc = H5Gcreate2(g1, "g2", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  }
  if (strstr(var, "dense")) {  /* more links than the compact limit (8): the root group switches to a fractal heap */
    /* "densebig": enough links for a root indirect block and internal B-tree nodes; "dense_in_group": the dataset's own
     * group (not the root) is the dense one */
    const int nlinks = strstr(var, "densebig") ? 3000 : 12;
    if (strstr(var, "dense_in_group")) {
      hid_t g1 = H5Gcreate2(f, "g1", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
      loc = H5Gcreate2(g1, "g2", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    }
    for (int i = 0; i < nlinks; ++i) {
      char nm[32];
      snprintf(nm, sizeof nm, i % 3 ? "other%02d" : "a_rather_longer_name_of_another_object_%05d", i);
      if (nlinks > 100) {  /* groups: cheaper than datasets */
        hid_t gg = H5Gcreate2(loc, nm, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        H5Gclose(gg);
      } else {
        hsize_t one = 1;
        hid_t sp = H5Screate_simple(1, &one, NULL);
        hid_t ds = H5Dcreate2(loc, nm, H5T_NATIVE_INT, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        H5Dclose(ds); H5Sclose(sp);
      }
    }
  }
  hid_t ftype = H5T_STD_U8LE, mtype = H5T_NATIVE_DOUBLE;
  if (!strcmp(dt, "u16")) ftype = H5T_STD_U16BE;
  else if (!strcmp(dt, "f32")) ftype = H5T_IEEE_F32LE;
  hid_t space = H5Screate_simple(3, dims, maxd);
  hid_t ds = H5Dcreate2(loc, name, ftype, space, H5P_DEFAULT, dcpl, H5P_DEFAULT);
  if (ds < 0) { fprintf(stderr, "H5Dcreate2 failed\n"); return 1; }
  const size_t n = (size_t)(dims[0] * dims[1] * dims[2]);
  double* buf = (double*)malloc(n * sizeof(double));
  for (hsize_t z = 0; z < dims[0]; ++z)
    for (hsize_t y = 0; y < dims[1]; ++y)
      for (hsize_t x = 0; x < dims[2]; ++x) buf[(z * dims[1] + y) * dims[2] + x] = value(z, y, x);
  if (strstr(var, "partial")) {  /* write only the planes [2, 5): the other chunks stay unallocated (fill value 0) */
    hsize_t start[3] = {2, 0, 0}, cnt[3] = {3, dims[1], dims[2]};
    hid_t ms = H5Screate_simple(3, cnt, NULL);
    H5Sselect_hyperslab(space, H5S_SELECT_SET, start, NULL, cnt, NULL);
    if (H5Dwrite(ds, mtype, ms, space, H5P_DEFAULT, buf + 2 * dims[1] * dims[2]) < 0) return 1;
    H5Sclose(ms);
  } else if (H5Dwrite(ds, mtype, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf) < 0) {
    return 1;
  }
  free(buf);
  H5Dclose(ds); H5Sclose(space); H5Pclose(dcpl); H5Fclose(f); H5Pclose(fapl);
  return 0;
}
