/* Driving the score-test engine from plain C through include/crm_hip.h -- what a non-Python host
 * (or the ctypes stub of INTEGRATION.md) does, call for call:
 *
 *   ctx -> background (mode B: E1 E1' + hK hK', decomposed on the device over the rho grid)
 *       -> gene (y, W, E0) -> panel (G) -> crm_scan_interaction -> p-values
 *
 * Input: one binary file of float64, header {n, k0, m, c, p} then y[n], E[n*k0], hK[n*m], W[n*c],
 * G[n*p] (all row-major).  Output: one line per variant "pvalue rho1" on stdout.
 *
 *   gcc -O2 -Iinclude examples/scan_from_c.c -Lcellregmap_amd -lcrm_hip -Wl,-rpath,$PWD/cellregmap_amd -o scan_from_c
 */
#include <stdio.h>
#include <stdlib.h>

#include "crm_hip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != CRM_OK) {                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, crm_last_error());     \
            return 1;                                                            \
        }                                                                        \
    } while (0)

static double* read_block(FILE* f, size_t count) {
    double* p = (double*)malloc(sizeof(double) * (count ? count : 1));
    if (!p || fread(p, sizeof(double), count, f) != count) {
        fprintf(stderr, "short read\n");
        exit(2);
    }
    return p;
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s cohort.bin\n", argv[0]);
        return 2;
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f) {
        perror(argv[1]);
        return 2;
    }
    double* hdr = read_block(f, 5);
    const long n = (long)hdr[0], p = (long)hdr[4];
    const int k0 = (int)hdr[1], m = (int)hdr[2], c = (int)hdr[3];
    double* y = read_block(f, (size_t)n);
    double* E = read_block(f, (size_t)n * k0);
    double* hK = read_block(f, (size_t)n * m);
    double* W = read_block(f, (size_t)n * c);
    double* G = read_block(f, (size_t)n * p);
    fclose(f);

    double rho[11];
    for (int i = 0; i < 11; i++) rho[i] = i / 10.0;

    crm_ctx* ctx = NULL;
    crm_background* bg = NULL;
    crm_gene* gene = NULL;
    crm_panel* panel = NULL;
    CHECK(crm_ctx_create(0, &ctx));
    CHECK(crm_background_create(ctx, n, E, k0, hK, m, 11, rho, 0.0, &bg));   /* CellRegMap(..., hK=hK) */
    CHECK(crm_gene_create(bg, y, W, c, E, k0, &gene));
    CHECK(crm_panel_create(ctx, n, G, p, p, &panel));
    double* pv = (double*)malloc(sizeof(double) * (size_t)(p ? p : 1));
    double* rho1 = (double*)malloc(sizeof(double) * (size_t)(p ? p : 1));
    CHECK(crm_scan_interaction(gene, panel, 0, p, NULL, NULL, pv, rho1, NULL, NULL, NULL, NULL, NULL, NULL, NULL,
                               NULL, NULL));
    for (long i = 0; i < p; i++) printf("%.17g %.17g\n", pv[i], rho1[i]);

    crm_panel_destroy(panel);
    crm_gene_destroy(gene);
    crm_background_destroy(bg);
    crm_ctx_destroy(ctx);
    free(pv); free(rho1); free(y); free(E); free(hK); free(W); free(G); free(hdr);
    return 0;
}
