// The host side of the fused step: qgd_setup.cpp buildFusedBlocks (CPU only; compiled by tests/test_fused_blocks_host.py with plain g++ from the
// library's own host sources).  Checks the tables the kernel trusts blindly:
//   * every owned cell is an own cell of exactly one block, ghost cells of none; every cell a neighbouring shard waits for sits in one of
//     the first nLayerBlocks blocks (whole bricks: they hold other cells too), and every one of those blocks holds such a cell;
//   * a block's face list holds every internal face of its own cells, once, with owner / neighbour / vertex positions that point at those very
//     labels; every list is padded to its stride with its last entry;
//   * the face entries of an own cell are the cell's faces in ascending label with the right side bit, patch faces as ~label;
//   * a vertex's cell positions are its pointCells in order, with the weights of the vertex kernel's table; patch points have count 0;
//   * the maxima are the maxima, and no block of more than 32 cells needs more LDS than lets three blocks share a CU.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>

#include "qgd_mesh.hpp"
#include "qgd_setup.hpp"

using namespace qgd;

static int fails = 0;
#define CHECK(cond, ...)                                   \
    do {                                                   \
        if (!(cond)) {                                     \
            if (fails < 20) { std::printf("FAILED %s:%d  %s  ", __FILE__, __LINE__, #cond); std::printf(__VA_ARGS__); std::printf("\n"); } \
            ++fails;                                       \
        }                                                  \
    } while (0)

static void checkMesh(const char* tag, HostMesh& m, double minMeanCells = 0.0, int maxTemplates = -1) {
    if (m.magSf.empty()) m.computeGeometry();
    m.computeDerived();
    const StaticData s = buildStaticData(m);
    const FusedBlocks B = buildFusedBlocks(s);
    std::printf("%s: %d cells, %d blocks (%d boundary-layer), strides %d %d %d %d %d, faces computed %lld of %d\n", tag, s.nC, B.nBlocks, B.nLayerBlocks, B.capC,
                B.capV, B.capF, B.capE, B.capPE, (long long)B.facesComputed, s.nIF);
    CHECK(B.nBlocks > 0, "%s: no blocks", tag);
    if (B.nBlocks == 0) return;
    {
        int64_t owned = 0;
        for (int32_t c = 0; c < s.nC; ++c) owned += (s.ghost.empty() || s.ghost[c] != 1) ? 1 : 0;
        const double mean = (double)owned / B.nBlocks;
        std::printf("    %.1f cells per block, %.2f cell records staged per cell, %.2f faces computed per cell\n", mean, (double)B.cellsStaged / owned,
                    (double)B.facesComputed / owned);
        CHECK(mean >= minMeanCells, "%s: %.1f cells per block on average, %.1f wanted", tag, mean, minMeanCells);
    }
    std::vector<int> ownerBlock((size_t)s.nC, -1);
    std::vector<int> tplUsed((size_t)std::max(B.nTemplates, 1), 0);
    std::printf("    %d templates for %d blocks (%s)\n", B.nTemplates, B.nBlocks, B.templated ? "shared" : "one per block");
    CHECK(maxTemplates < 0 || B.nTemplates <= maxTemplates, "%s: %d templates, at most %d wanted", tag, B.nTemplates, maxTemplates);
    int32_t maxTot = 0, maxAll = 0, maxV = 0, maxF = 0, maxLds = 0;
    int64_t faces = 0;
    for (int32_t b = 0; b < B.nBlocks; ++b) {
        const int32_t nOwn = B.hdr[4 * b], nAll = B.hdr[4 * b + 1], nV = B.hdr[4 * b + 2], nF = B.hdr[4 * b + 3], nTot = B.hdr2[4 * b];
        CHECK(nOwn >= 1 && nOwn <= kFusedCells && nOwn <= nAll && nAll <= nTot && nTot <= B.capC && nAll <= kFusedCapC && nTot <= kFusedCapTot, "block %d counts", b);
        CHECK(nV <= B.capV && nV <= kFusedCapV && nF <= B.capF && nF <= kFusedCapF, "block %d counts", b);
        maxTot = std::max(maxTot, nTot); maxAll = std::max(maxAll, nAll); maxV = std::max(maxV, nV); maxF = std::max(maxF, nF);
        faces += nF;
        const int32_t lds = 48 * nTot + 32 * nAll + std::max(72 * nV + 24 * nAll, 40 * nF);   // the kernel's layout of this block's records
        maxLds = std::max(maxLds, lds);
        CHECK(lds <= kFusedLdsTarget || nOwn <= 32, "block %d needs %d B of LDS with %d own cells", b, lds, nOwn);
        const int32_t* cells = &B.cells[(size_t)b * B.capC];
        const int32_t* verts = &B.verts[(size_t)b * B.capV];
        for (int32_t i = nTot; i < B.capC; ++i) CHECK(cells[i] == cells[nTot - 1], "block %d cell padding", b);
        for (int32_t i = std::max(nV, 1); i < B.capV; ++i) CHECK(verts[i] == verts[std::max(nV, 1) - 1], "block %d vertex padding", b);
        std::set<int32_t> distinct(cells, cells + nTot);
        CHECK((int32_t)distinct.size() == nTot, "block %d lists a cell twice", b);
        bool hasLayerCell = false;
        for (int32_t j = 0; j < nOwn; ++j) {
            const int32_t c = cells[j];
            const int role = s.ghost.empty() ? 0 : s.ghost[c];
            CHECK(role != 1, "ghost cell %d owned by block %d", c, b);
            CHECK(ownerBlock[c] == -1, "cell %d owned twice", c);
            ownerBlock[c] = b;
            CHECK(role != 2 || b < B.nLayerBlocks, "cell %d (role %d) in block %d of %d boundary-layer blocks", c, role, b, B.nLayerBlocks);
            if (role == 2) hasLayerCell = true;
        }
        CHECK(hasLayerCell || b >= B.nLayerBlocks || nOwn < 1, "block %d of the %d boundary-layer blocks holds no cell a neighbour waits for", b, B.nLayerBlocks);
        // faces
        std::set<int32_t> want;
        for (int32_t j = 0; j < nOwn; ++j) {
            const int32_t c = cells[j];
            const size_t base = (size_t)s.cfSlice[c >> 6] * 64 + (c & 63);
            for (int i = 0; i < s.cfCount[c]; ++i) {
                const int32_t it = s.cfItem[base + (size_t)i * 64], f = it >= 0 ? it : ~it;
                if (f < s.nIF) want.insert(f);
            }
        }
        CHECK((int32_t)want.size() == nF, "block %d: %d faces listed, %d wanted", b, nF, (int)want.size());
        // the block's local topology sits in its template (hdr2[1]); blocks that share one must expand to their own labels all the same
        const int32_t tpl = B.hdr2[4 * b + 1];
        CHECK(tpl >= 0 && tpl < B.nTemplates && (B.templated || tpl == b) && (B.templated == (B.nTemplates < B.nBlocks)), "block %d template %d of %d", b, tpl, B.nTemplates);
        tplUsed[tpl]++;
        const int32_t* face = &B.faceLabel[(size_t)b * B.capF];
        const uint32_t* fpos = &B.facePos[(size_t)tpl * B.capF * 3];
        for (int32_t lf = 0; lf < nF; ++lf) {
            const int32_t f = face[lf];
            const uint32_t lc = fpos[3 * lf], va = fpos[3 * lf + 1], vb = fpos[3 * lf + 2];
            CHECK(want.count(f) == 1, "block %d lists face %d", b, f);
            CHECK((int32_t)(lc & 0xffff) < nAll && (int32_t)(lc >> 16) < nAll, "block %d face %d cell positions", b, f);
            CHECK(cells[lc & 0xffff] == s.own[f] && cells[lc >> 16] == s.nei[f], "block %d face %d owner / neighbour", b, f);
            const uint32_t pv[4] = {va & 0xffff, va >> 16, vb & 0xffff, vb >> 16};
            for (int q = 0; q < 4; ++q) {
                const int32_t v = s.verts[4 * (size_t)f + q];
                if (v >= 0) CHECK((int32_t)pv[q] < nV && verts[pv[q]] == v, "block %d face %d vertex %d", b, f, q);
            }
            if (lf > 0) CHECK(face[lf - 1] < f, "block %d faces not ascending", b);
        }
        for (int32_t lf = nF; lf < B.capF; ++lf) {
            CHECK(nF == 0 || face[lf] == face[nF - 1], "block %d face padding", b);
            for (int q = 0; q < 3; ++q) CHECK(nF == 0 || fpos[3 * lf + q] == fpos[3 * (nF - 1) + q], "block %d face padding", b);
        }
        // face entries of the own cells
        for (int32_t j = 0; j < kFusedCells; ++j) {
            const int nE = B.nEntry[(size_t)b * kFusedCells + j];
            if (j >= nOwn) { CHECK(nE == 0, "block %d entry count beyond the own cells", b); continue; }
            const int32_t c = cells[j];
            CHECK(nE == s.cfCount[c] && nE <= B.capE, "block %d cell %d entry count", b, c);
            const size_t base = (size_t)s.cfSlice[c >> 6] * 64 + (c & 63);
            for (int e = 0; e < nE; ++e) {
                const int32_t it = s.cfItem[base + (size_t)e * 64], f = it >= 0 ? it : ~it;
                const int32_t got = B.entry[((size_t)tpl * B.capE + e) * kFusedCells + j];
                if (f >= s.nIF) CHECK(got == ~f && it >= 0, "block %d cell %d patch-face entry", b, c);
                else CHECK(got >= 0 && (got >> 1) < nF && face[got >> 1] == f && (got & 1) == (it < 0 ? 1 : 0), "block %d cell %d entry %d", b, c, e);
            }
        }
        // vertex tables
        for (int32_t lv = 0; lv < B.capV; ++lv) {
            const int n = B.vCount[(size_t)b * B.capV + lv];
            if (lv >= nV) { CHECK(n == 0, "block %d vertex count beyond the list", b); continue; }
            const int32_t v = verts[lv];
            CHECK(n == s.pcCount[v] && n <= B.capPE, "block %d vertex %d count", b, v);
            const size_t base = (size_t)s.pcSlice[v >> 6] * 64 + (v & 63);
            for (int e = 0; e < n; ++e) {
                const int32_t pos = B.vPos[((size_t)tpl * B.capPE + e) * B.capV + lv];
                CHECK(pos < nTot && cells[pos] == s.pcCell[base + (size_t)e * 64], "block %d vertex %d cell %d", b, v, e);
                CHECK(B.vW[((size_t)b * B.capPE + e) * B.capV + lv] == s.pcW[base + (size_t)e * 64], "block %d vertex %d weight %d", b, v, e);
            }
        }
    }
    for (int32_t c = 0; c < s.nC; ++c) {
        const int role = s.ghost.empty() ? 0 : s.ghost[c];
        CHECK((ownerBlock[c] >= 0) == (role != 1), "cell %d (role %d) block %d", c, role, ownerBlock[c]);
    }
    for (int32_t t = 0; t < B.nTemplates; ++t) CHECK(tplUsed[t] >= 1, "%s: template %d is nobody's", tag, t);
    CHECK(maxTot == B.maxTot && maxAll == B.maxAll && maxV == B.maxV && maxF == B.maxF, "%s: maxima %d %d %d %d vs %d %d %d %d", tag, maxTot, maxAll, maxV,
          maxF, B.maxTot, B.maxAll, B.maxV, B.maxF);
    CHECK(maxLds == B.maxLds, "%s: LDS %d vs %d", tag, maxLds, B.maxLds);
    CHECK(faces == B.facesComputed && faces >= 0, "%s: faces computed", tag);
}

int main() {
    const double lo[3] = {0, 0, 0}, hi[3] = {1, 1, 1};
    const int32_t pt[6] = {0, 0, 0, 0, 0, 0};
    { HostMesh m = makeBox(16, 8, 8, 0, 8, lo, hi, pt); checkMesh("box 16x8x8 (bricks)", m); }
    { HostMesh m = makeBox(13, 7, 5, 0, 5, lo, hi, pt); checkMesh("box 13x7x5 (ragged)", m); }
    { HostMesh m = makeBox(3, 2, 2, 0, 2, lo, hi, pt); checkMesh("box 3x2x2", m); }
    {
        HostMesh m = makeBox(11, 9, 7, 0, 7, lo, hi, pt);
        jitterPoints(m, 0.15, 7);
        splitQuads(m, 3);
        splitEdges(m, 5);
        checkMesh("jittered, every third quad split, every fifth edge split (triangles, polygons)", m);
    }
    { HostMesh m = makeBox(12, 6, 12, 3, 9, lo, hi, pt); checkMesh("slab 3..9 of a 12x6x12 box (two cuts)", m); }
    // what a cut must not cost: a 50-plane slab between two cuts keeps brick-shaped blocks -- 13 brick layers of 3 or 4
    // planes, 123 cells per block -- and the blocks of the planes a neighbour waits for are whole bricks, not flat one-plane ones
    { HostMesh m = makeBox(80, 80, 150, 49, 101, lo, hi, pt); checkMesh("planes 50..100 of an 80x80x150 box (50 owned planes between two ghost planes)", m, 120.0, 2600 - 8 * 18 * 11 + 40); }   // the 8 x 18 x 11 bricks without a patch face or a ghost plane share a handful of templates
    { HostMesh m = makeBox(50, 50, 50, 0, 50, lo, hi, pt); checkMesh("box 50x50x50 (extents 8x4x4 bricks do not divide: 5x5x5 ones)", m, 120.0, 1000 - 8 * 8 * 8 + 1); }   // 512 interior cubes, ONE template
    {   // a jittered, Morton-renumbered mesh: no two blocks list their cells alike -> one template per block, the kernel's untemplated path
        HostMesh m = makeBox(24, 16, 16, 0, 16, lo, hi, pt);
        jitterPoints(m, 0.15, 7);
        splitQuads(m, 7);
        checkMesh("jittered 24x16x16, every seventh quad split", m, 0.0, -1); }
    if (fails) { std::printf("%d checks failed\n", fails); return 1; }
    std::printf("ok\n");
    return 0;
}
