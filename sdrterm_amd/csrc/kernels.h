// Kernel argument blocks and launch helpers (kernels.hip <-> api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "../../include/sdrterm_hip.h"
#include "tables.h"

#define SDRT_MAX_OUT_SECTIONS 4

struct DecimArgs {
    const void *raw;            // [n_chunks][n][2] records of the input encoding
    const DecimTables *tab;     // device copy
    const double *nco_tile;     // [rows][16*L][2]
    const double *nco_phase;    // [rows][n_tiles][2]
    double *mid;                // FM: [row][chunk][m/2] pair angles; AM: [row][chunk][m] |y|^2
    double *ydec;               // optional [row][chunk][m][2]
    size_t n_chunks;
    int n, m, n_tiles, rows, mode, swap;
    double norm_scale, norm_bias;   // re' = re*scale + bias, im' = im*scale (identity: 1, 0)
};

struct PostArgs {
    const double *mid;
    const double *resample_odd;  // [m/2] (FM)
    const double *out_sos;       // [out_sections][6]
    double *out;                 // [row][chunk][m]
    int m, mode, out_sections;
};

size_t decim_lds_bytes(int n, int L, int m, bool nco);
size_t post_lds_bytes(int m, int mode);
hipError_t launch_decim(const DecimArgs &a, int L, int dt, bool nco, size_t n_chunks, hipStream_t s);
hipError_t launch_post(const PostArgs &a, size_t items, hipStream_t s);
hipError_t launch_mfma_selftest(const double *A, const double *B, double *D, hipStream_t s);
