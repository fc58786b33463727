/* Stand-alone C caller of libdavo_hip.so: no Python, no HIP headers, plain pointers and sizes.
 *
 *   gcc -O2 -I include -o c_abi_pose examples/c_abi_pose.c -L davo_amd -ldavo_hip -Wl,-rpath,$PWD/davo_amd -lm
 *   ./c_abi_pose weights.bin inputs.bin poses_out.bin
 *
 * weights.bin: repeated records  [u32 name_len][name][u32 rank][i64 dims[rank]][f32 data]   (TF variable names)
 * inputs.bin : [i32 B][i32 H][i32 W] then img u8 [B,H,3W,3], flow f32 [B,4,H,W,2], seg f32 [B,3,H,W,1]
 * poses_out  : f32 [B,2,6]
 * tests/test_hip_parity.py::test_c_caller writes the two input files, runs this program and compares the
 * poses with the Python host class bit for bit (both are thin layers over the same entry points). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "davo_hip.h"

static void die(davo_ctx* ctx, const char* what) {
    fprintf(stderr, "%s: %s\n", what, ctx ? davo_last_error(ctx) : "no context");
    exit(1);
}

static void* slurp(FILE* f, size_t n) {
    void* p = malloc(n ? n : 1);
    if (!p || fread(p, 1, n, f) != n) { fprintf(stderr, "short read (%zu bytes)\n", n); exit(1); }
    return p;
}

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s weights.bin inputs.bin poses_out.bin\n", argv[0]); return 2; }
    FILE* fi = fopen(argv[2], "rb");
    if (!fi) { perror(argv[2]); return 1; }
    int32_t hdr[3];
    if (fread(hdr, sizeof hdr, 1, fi) != 1) { fprintf(stderr, "bad inputs header\n"); return 1; }
    const int B = hdr[0], H = hdr[1], W = hdr[2];
    const size_t px = (size_t)H * W;
    uint8_t* img = slurp(fi, (size_t)B * px * 9);
    float* flow = slurp(fi, (size_t)B * px * 8 * sizeof(float));
    float* seg = slurp(fi, (size_t)B * px * 3 * sizeof(float));
    fclose(fi);

    /* the flagship --version, as davo_amd/version.py:parse_version derives it */
    const davo_variant v = {5, 128, 1, 0, 3, 1, 1, 1};
    davo_ctx* ctx = NULL;
    if (davo_create(&ctx, 0, H, W, B, &v) != DAVO_OK) die(ctx, "davo_create");

    FILE* fw = fopen(argv[1], "rb");
    if (!fw) { perror(argv[1]); return 1; }
    uint32_t nlen;
    while (fread(&nlen, sizeof nlen, 1, fw) == 1) {
        char name[256];
        uint32_t rank;
        int64_t dims[8];
        if (nlen >= sizeof name || fread(name, 1, nlen, fw) != nlen || fread(&rank, sizeof rank, 1, fw) != 1 || rank > 8 ||
            fread(dims, sizeof(int64_t), rank, fw) != rank) { fprintf(stderr, "bad weight record\n"); return 1; }
        name[nlen] = 0;
        size_t n = 1;
        for (uint32_t i = 0; i < rank; ++i) n *= (size_t)dims[i];
        float* data = slurp(fw, n * sizeof(float));
        if (davo_load_weight(ctx, name, data, dims, (int)rank) != DAVO_OK) die(ctx, name);
        free(data);
    }
    fclose(fw);
    if (davo_weights_missing(ctx) != 0) die(ctx, "weights missing");

    float* poses = malloc((size_t)B * 12 * sizeof(float));
    if (davo_forward(ctx, B, img, flow, seg, poses) != DAVO_OK) die(ctx, "davo_forward");
    FILE* fo = fopen(argv[3], "wb");
    if (!fo || fwrite(poses, sizeof(float), (size_t)B * 12, fo) != (size_t)B * 12) { perror(argv[3]); return 1; }
    fclose(fo);
    printf("window 0, tgt->src0: rz ry rx tx ty tz = %g %g %g %g %g %g\n", poses[0], poses[1], poses[2], poses[3], poses[4], poses[5]);
    davo_destroy(ctx);
    free(poses); free(img); free(flow); free(seg);
    return 0;
}
