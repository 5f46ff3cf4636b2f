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
    const OutTables *otab;      // device copy
    const double *nco_tile;     // [rows][L][2]     exp(i theta j), j < L            (sample within block)
    const double *nco_phase;    // [rows][n/L][2]   exp(i theta L B), B < n/L        (block within chunk)
    const double *resample_odd; // [m/2] (FM)
    const double *out_sos;      // [out_sections][6], used only by the sequential fallback
    double *out;                // [row][chunk][m]
    double *ydec;               // optional [row][chunk][m][2]
    unsigned long long *stamps; // optional diagnostic: [wg < stamp_wgs][wave][8] s_memtime at phase boundaries
    int stamp_wgs;
    size_t n_chunks;
    int n, m, n_tiles, rows, mode, swap, out_sections;
    double norm_scale, norm_bias;   // re' = re*scale + bias, im' = im*scale (identity: 1, 0)
};

size_t chain_lds_bytes(int n, int L, int m, bool nco, bool fm);
hipError_t launch_chain(const DecimArgs &a, int L, int dt, bool nco, size_t n_chunks, hipStream_t s);
hipError_t launch_mfma_selftest(const double *A, const double *B, double *D, hipStream_t s);
