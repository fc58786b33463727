// plan.hip — launch planning (see plan.h).  Host logic only.
#include "plan.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace davo {

#ifdef DAVO_TUNING
const char* tuning_env(const char* name) { return getenv(name); }
#endif

// ---- FP32-MFMA path -------------------------------------------------------------------------------
// Workgroups of one launch all take the same time, and the dispatcher refills both slots of a
// CU together, so a grid that is not a whole number of rounds (256 CUs x resident workgroups)
// pays for a full last round: at B = 32 the 3,328 tiles of cnv5/cnv6 are 6.5 rounds of 512 and
// ran in the time of 7.  A layer is therefore issued as a main launch of whole rounds at the
// widest N tile plus, when it pays, a remainder launch with a narrower N tile (more, shorter
// workgroups) that again fills whole rounds.  Costs are in units of one round of 128x128 tiles.
static int slots_for(int BN, int ncu) { return ncu * (BN == 32 ? 3 : 2); }         // LDS 46 / 55 / 74 KB per workgroup
// time of one round (every CU full) relative to a round of 128x128 tiles: resident workgroups
// per CU x tile area / measured relative efficiency of the narrower tiles
static double tile_cost(int BN) { return BN == 128 ? 1.0 : BN == 64 ? 0.5 / 0.92 : 0.375 / 0.75; }

static double rounds_cost(long tiles, int BN, int ncu) {
    const long s = slots_for(BN, ncu);
    return (double)((tiles + s - 1) / s) * tile_cost(BN);
}

std::vector<Launch> plan_layer(int mtiles, int npad, int groups, int ncu) {
    int bmax = npad % 128 == 0 ? 128 : npad % 64 == 0 ? 64 : 32;
    // tuning build only: DAVO_FORCE_BN=32|64|128 -> one launch at that N tile, DAVO_PLAN=single -> one launch
    // at the widest N tile
    if (const char* e = tuning_env("DAVO_FORCE_BN")) {
        const int bn = atoi(e);
        if ((bn == 32 || bn == 64 || bn == 128) && npad % bn == 0) return {{0, mtiles, bn}};
    }
    if (const char* e = tuning_env("DAVO_PLAN"))
        if (!strcmp(e, "single")) return {{0, mtiles, bmax}};
    std::vector<Launch> best;
    double best_cost = 1e30;
    for (int bn = bmax; bn >= 32; bn >>= 1) {                          // one launch
        const double c = rounds_cost((long)mtiles * (npad / bn) * groups, bn, ncu);
        if (c < best_cost - 1e-9) { best_cost = c; best = {{0, mtiles, bn}}; }
    }
    const long per_m = (long)(npad / bmax) * groups;                   // tiles per M tile at bmax
    const long s = slots_for(bmax, ncu);
    const int main_m = (int)(((long)mtiles * per_m / s) * s / per_m);  // whole rounds only
    if (main_m > 0 && main_m < mtiles && (main_m * per_m) % s == 0) {
        const double cm = rounds_cost(main_m * per_m, bmax, ncu);
        for (int bn = bmax; bn >= 32; bn >>= 1) {
            const double c = cm + rounds_cost((long)(mtiles - main_m) * (npad / bn) * groups, bn, ncu) + 0.02;
            if (c < best_cost - 1e-9) { best_cost = c; best = {{0, main_m, bmax}, {main_m, mtiles - main_m, bn}}; }
        }
    }
    return best;
}

// ---- f16x3 path -----------------------------------------------------------------------------------
TileShape tile_shape(int t) {
    switch (t) {
        case TILE_128x32: return {128, 32, 256, TileH<4, 1, 1, 1>::LDS_BYTES_DMA};
        case TILE_256x64: return {256, 64, 512, TileH<4, 2, 2, 1>::LDS_BYTES_DMA};
        case TILE_256x128: return {256, 128, 512, TileH<4, 2, 2, 2>::LDS_BYTES_DMA};
        case TILE_128x256: return {128, 256, 512, TileH<2, 4, 2, 2>::LDS_BYTES_DMA};
        case TILE_256x256: return {256, 256, 512, TileH<4, 2, 2, 4>::LDS_BYTES_DMA};
        case TILE_208x256: return {TileS::BM, TileS::BN, TileS::THREADS, TileS::LDS_BYTES};
        case TILE_208x128: return {TileSW<4>::BM, TileSW<4>::BN, TileSW<4>::THREADS, TileSW<4>::lds_bytes(3)};
        default: return {128, 128, 512, TileH<4, 2, 1, 2>::LDS_BYTES_DMA};
    }
}

// Same idea as plan_layer: whole rounds of the most efficient tile, then a remainder launch with a
// smaller tile that again fills whole rounds.  Costs are in units of one round of 256x256 tiles;
// eff = measured throughput of the tile relative to 256x256 at full occupancy (B=128: cnv5, cnv6, cnv7 forced to
// one tile shape each, re-measured after the matrix loop was software-pipelined).  The two narrow tiles keep the
// figures fitted on the K < 600 layers that use them (cnv3, cnv4: A/B on one box, profiles/ r01f notes).
// The 208x256 tile (conv_igemm_h3s.h) is offered only to the layers it is instantiated for and only as a single launch;
// 0.93: cnv6 at B=128 (16 whole rounds) 1.995 ms against 1.869 ms on 13 rounds of 256x256, cnv5 1.071 against 0.997 (three pixel
// ring slots on the 208 tile, shared-tap staging on the 256 tile).
// Round 2: 256x256 and 256x128 stage one pixel patch per filter row for its three taps on cnv3..cnv6 (-3 % and -6 %:
// gpurun_out/ab_r02v.log, ab_r02w.log), which moves 256x128 past 128x128 where no round is left half empty (cnv4 at B=128).
static TileInfo kTiles[] = {{TILE_256x256, 1, 1.00}, {TILE_128x256, 1, 0.82}, {TILE_256x128, 1, 0.89},
                            {TILE_128x128, 2, 0.87}, {TILE_256x64, 2, 0.62}, {TILE_128x32, 4, 0.40}, {TILE_208x256, 1, 0.93},
                            {TILE_208x128, 1, 0.75}};      // 208x128: cnv4 only, offered with "tile_208x128" 1; fitted on cnv4 at B = 32 (profiles/r04_cnv4_208x128_ab.log)

// tuning build only: DAVO_H3_EFF="e0,e1,e2,e3,e4,e5[,p4]" (and DAVO_H3_EFF208=e) overrides the efficiencies (table order) and 256x64's per_cu
static void tiles_from_env() {
    static bool done = false;
    if (done) return;
    done = true;
    const char* e = tuning_env("DAVO_H3_EFF");
    if (!e) return;
    double v[7] = {0, 0, 0, 0, 0, 0, 0};
    const int n = sscanf(e, "%lf,%lf,%lf,%lf,%lf,%lf,%lf", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6);
    for (int i = 0; i < 6 && i < n; ++i) if (v[i] > 0) kTiles[i].eff = v[i];
    if (n >= 7 && v[6] >= 1) kTiles[4].per_cu = (int)v[6];
    if (const char* e2 = tuning_env("DAVO_H3_EFF208")) { const double x = atof(e2); if (x > 0) kTiles[6].eff = x; }
}

const TileInfo* h3_tiles(int* n) {
    tiles_from_env();
    if (n) *n = (int)(sizeof kTiles / sizeof kTiles[0]);
    return kTiles;
}

// whole rounds run per_cu workgroups per CU side by side; in the last, partial round a CU holds
// ceil(rest / 256) of them (the dispatcher spreads a short tail one per CU)
double h3_cost(const TileInfo& t, long ntiles, int ncu) {
    const TileShape ts = tile_shape(t.id);
    const long slots = (long)ncu * t.per_cu;
    const double one = (ts.bm * ts.bn / 65536.0) / t.eff;
    const long full = ntiles / slots, rest = ntiles % slots;
    return (double)full * t.per_cu * one + (double)((rest + ncu - 1) / ncu) * one;
}

std::vector<LaunchH> plan_layer_h3(int M, int npad, int groups, int forced_tile, bool allow_208, int ncu, double others_scale) {
    tiles_from_env();
    TileInfo tiles[sizeof kTiles / sizeof kTiles[0]];
    for (size_t i = 0; i < sizeof kTiles / sizeof kTiles[0]; ++i) {
        tiles[i] = kTiles[i];
        if (tiles[i].id != TILE_256x256) tiles[i].eff *= others_scale;
    }
    auto ntiles = [&](const TileInfo& t, int rows) {
        const TileShape ts = tile_shape(t.id);
        return (long)((rows + ts.bm - 1) / ts.bm) * (npad / ts.bn) * groups;
    };
    auto fits = [&](const TileInfo& t) {
        const int bn = tile_shape(t.id).bn;
        return bn <= npad && npad % bn == 0 && (!is_208(t.id) || (allow_208 && bn == npad));
    };
    if (forced_tile >= 0 && forced_tile != TILE_MERGED_MARK && (!is_208(forced_tile) || (allow_208 && tile_shape(forced_tile).bn == npad))) return {{0, M, forced_tile}};
    std::vector<LaunchH> best;
    double best_cost = 1e30;
    const char* rf = tuning_env("DAVO_H3_REM_TILE");
    const int rem_force = rf ? atoi(rf) : -1;
    for (const TileInfo& t1 : tiles) {
        if (!fits(t1)) continue;
        const double c1 = h3_cost(t1, ntiles(t1, M), ncu);
        if (c1 < best_cost - 1e-9) { best_cost = c1; best = {{0, M, t1.id}}; }
        if (is_208(t1.id)) continue;                                       // single launch only
        const TileShape s1 = tile_shape(t1.id);
        const long per_round = (long)ncu * t1.per_cu, per_m = (long)(npad / s1.bn) * groups;
        if (per_round % per_m) continue;
        const long m_per_round = per_round / per_m;                       // M tiles of t1 in one round
        const long rounds = ((long)M / s1.bm) / m_per_round;
        int rows1 = (int)(rounds * m_per_round * s1.bm);
        rows1 -= rows1 % 256;                                             // every tile height divides 256
        if (rows1 <= 0 || rows1 >= M) continue;
        const double cm = h3_cost(t1, ntiles(t1, rows1), ncu);
        for (const TileInfo& t2 : tiles) {
            if (!fits(t2) || is_208(t2.id)) continue;
            if (rem_force >= 0 && t2.id != rem_force) continue;
            const double c = cm + h3_cost(t2, ntiles(t2, M - rows1), ncu) + 0.08;      // a second launch: its fill/drain and the kernel boundary
            if (c < best_cost - 1e-9) { best_cost = c; best = {{0, rows1, t1.id}, {rows1, M - rows1, t2.id}}; }
        }
    }
    return best;
}

int plan_single_tile_h3(int M, int npad, int groups, int max_bm, int forced_tile, bool allow_208, int ncu) {
    tiles_from_env();
    for (int pass = 0; pass < 2; ++pass) {                               // a forced tile that does not fit is ignored
        int best = -1;
        double bc = 1e30;
        for (const TileInfo& t : kTiles) {
            const TileShape ts = tile_shape(t.id);
            if (ts.bn > npad || npad % ts.bn || ts.bm > max_bm || (is_208(t.id) && !(allow_208 && ts.bn == npad))) continue;
            if (pass == 0 && forced_tile >= 0 && t.id != forced_tile) continue;
            const double cst = h3_cost(t, (long)((M + ts.bm - 1) / ts.bm) * (npad / ts.bn) * groups, ncu);
            if (cst < bc) { bc = cst; best = t.id; }
        }
        if (best >= 0 || forced_tile < 0) return best;
    }
    return -1;
}

}  // namespace davo
