// plan.h — launch planning of the convolution layers: which tile shape, and how a layer is split
// into a main launch of whole rounds plus a remainder launch.  Pure host logic (plan.hip); the
// test hook davo_plan_layer (include/davo_hip.h) exposes it to the CPU test-suite.
#pragma once
#include <vector>

#include "params.h"

namespace davo {

// Tuning knobs (environment variables, the measurement-only `dbg` bits of the kernels) exist only in a
// -DDAVO_TUNING build (tools/build_variant.py); the product library reads no environment.
#ifdef DAVO_TUNING
const char* tuning_env(const char* name);
#else
inline const char* tuning_env(const char*) { return nullptr; }
#endif

// ---- FP32-MFMA path (conv_igemm.h): 128-row M tiles x BN ---------------------------------------
struct Launch { int mtile0, mtiles, BN; };
// ncu: compute units the launch may use (256 = the whole MI355X; fewer when the context's stream is CU-masked)
std::vector<Launch> plan_layer(int mtiles, int npad, int groups, int ncu = 256);

// ---- f16x3 path (conv_igemm_h3.h) -----------------------------------------------------------------
// tile id -> (WM, WN, TM, TN): BM = WM*TM*32, BN = WN*TN*32, threads = WM*WN*64
// TILE_208x256 is conv_igemm_h3s.h (waves split the channels; whole rounds for the PoseNN's 16x208-pixel feature maps)
// 7 is not a tile: davo_last_plan reports it for the merged main + remainder grid.  TILE_208x128 (round 4): the four-wave conv_igemm_h3s for cnv4
enum { TILE_128x32 = 0, TILE_256x64 = 1, TILE_256x128 = 2, TILE_128x256 = 3, TILE_128x128 = 4, TILE_256x256 = 5, TILE_208x256 = 6, TILE_MERGED_MARK = 7, TILE_208x128 = 8, NUM_TILES = 9 };
inline bool is_208(int t) { return t == TILE_208x256 || t == TILE_208x128; }
struct TileShape { int bm, bn, threads, lds; };
TileShape tile_shape(int t);

struct TileInfo { int id, per_cu; double eff; };
struct LaunchH { int row0, rows, tile; };
double h3_cost(const TileInfo& t, long ntiles, int ncu = 256);
const TileInfo* h3_tiles(int* n);
// main launch + optional remainder launch for an M x npad (x groups) layer; forced_tile >= 0: one launch of that tile
// allow_208: the layer has a conv_igemm_h3s instantiation for its width (3x3, Cin >= 32: cnv5, cnv6, cnv7 at 256 output channels
// per group -> TILE_208x256; cnv4 at 128 -> TILE_208x128)
// others_scale: the efficiencies of every tile but 256x256 are multiplied by it (cnv5 / cnv6 with "wave128": their 256x256 launches run
// on conv_igemm_h3w, 4.7 % faster than the kernel the table was fitted on)
std::vector<LaunchH> plan_layer_h3(int M, int npad, int groups, int forced_tile, bool allow_208 = false, int ncu = 256, double others_scale = 1.0);
// one launch, one tile shape no taller than max_bm rows (fused pose head: a tile touches <= 2 images); -1 if none fits
int plan_single_tile_h3(int M, int npad, int groups, int max_bm, int forced_tile, bool allow_208 = false, int ncu = 256);

}  // namespace davo
