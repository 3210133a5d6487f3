// heat_system.out -- writes the steady 2-D heat problem of the reference's sibling task as a dense
// SPD linear system in the LAM file format, so the CG drivers can solve it in file mode
// (BASELINE.json configs[4]).
//
// Problem definition (and nothing else) is taken from
// /root/reference/heat_equation-main/src/heat_equation.cpp: an nx x ny grid stored [y][x] (:203),
// Dirichlet boundary north(y=ny-1)=0, south(y=0)=west=east=100 (:165-168, set_initial_solution
// :27-37), five-point average in the interior (:75-89).  The reference solves it by Jacobi sweeps
// until max_diff < 1e-3 (:130,164) and contains no matrix assembly; here the same discrete equations
//      4 T(x,y) - sum over INTERIOR neighbours T(nb) = sum over BOUNDARY neighbours T_bc(nb)
// are written for the (nx-2)(ny-2) interior unknowns, numbered k = (y-1)(nx-2) + (x-1).  The matrix
// is the 2-D Dirichlet Laplacian: symmetric positive definite, stored densely (n x n doubles).
//
//   heat_system.out assemble nx ny matrix.bin rhs.bin      write the system
//   heat_system.out field    nx ny sol.bin heat.bin        put a solution vector back on the grid,
//                                                           in the format the reference writes (:7-23)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace
{
const double BC_NORTH = 0.0, BC_SOUTH = 100.0, BC_WEST = 100.0, BC_EAST = 100.0;

// closes on every exit path (the early `return 2`s used to leak the handles)
struct File {
    FILE *f;
    File(const char *path, const char *mode) : f(fopen(path, mode)) {}
    ~File() { if (f) fclose(f); }
    File(const File &) = delete;
    File &operator=(const File &) = delete;
    operator FILE *() const { return f; }
    bool close() { const bool ok = f == nullptr || fclose(f) == 0; f = nullptr; return ok; }
};

bool write_header(FILE *f, uint64_t rows, uint64_t cols)
{
    const uint64_t h[2] = {rows, cols};
    return fwrite(h, sizeof(uint64_t), 2, f) == 2;
}

int assemble(size_t nx, size_t ny, const char *mpath, const char *bpath)
{
    if (nx < 3 || ny < 3) { fprintf(stderr, "need nx, ny >= 3\n"); return 1; }
    const size_t mx = nx - 2, my = ny - 2, n = mx * my;
    File fm(mpath, "wb"), fb(bpath, "wb");
    if (!fm.f || !fb.f) { fprintf(stderr, "Cannot open output file\n"); return 2; }
    if (!write_header(fm, n, n) || !write_header(fb, n, 1)) return 2;
    std::vector<double> row(n), rhs(n, 0.0);
    for (size_t y = 1; y <= my; y++)
        for (size_t x = 1; x <= mx; x++) {
            const size_t k = (y - 1) * mx + (x - 1);
            std::fill(row.begin(), row.end(), 0.0);
            row[k] = 4.0;
            if (y + 1 <= my) row[k + mx] = -1.0; else rhs[k] += BC_NORTH;   // y+1 == ny-1: north edge
            if (y - 1 >= 1) row[k - mx] = -1.0; else rhs[k] += BC_SOUTH;    // y-1 == 0: south edge
            if (x - 1 >= 1) row[k - 1] = -1.0; else rhs[k] += BC_WEST;
            if (x + 1 <= mx) row[k + 1] = -1.0; else rhs[k] += BC_EAST;
            if (fwrite(row.data(), sizeof(double), n, fm) != n) return 2;
        }
    if (fwrite(rhs.data(), sizeof(double), n, fb) != n) return 2;
    if (!fm.close() || !fb.close()) { fprintf(stderr, "Cannot write output file\n"); return 2; }
    printf("heat system: grid %zux%zu -> n=%zu unknowns, dense matrix %.3f GB\n", nx, ny, n, n * (double)n * 8 / 1e9);
    return 0;
}

int field(size_t nx, size_t ny, const char *spath, const char *hpath)
{
    const size_t mx = nx - 2, my = ny - 2, n = mx * my;
    File fs(spath, "rb");
    if (!fs.f) { fprintf(stderr, "Cannot open solution file\n"); return 2; }
    uint64_t h[2];
    std::vector<double> x(n);
    if (fread(h, sizeof(uint64_t), 2, fs) != 2 || h[0] != n || fread(x.data(), sizeof(double), n, fs) != n) {
        fprintf(stderr, "solution file does not hold %zu values\n", n);
        return 2;
    }
    fs.close();
    std::vector<double> heat(nx * ny, 0.0);
    for (size_t i = 1; i + 1 < nx; i++) { heat[(ny - 1) * nx + i] = BC_NORTH; heat[i] = BC_SOUTH; }
    for (size_t j = 1; j + 1 < ny; j++) { heat[j * nx] = BC_WEST; heat[j * nx + nx - 1] = BC_EAST; }
    heat[0] = (BC_SOUTH + BC_WEST) / 2;                         // corners as the reference sets them (:34-37);
    heat[(ny - 1) * nx] = (BC_NORTH + BC_WEST) / 2;             // they never enter the stencil
    heat[nx - 1] = (BC_SOUTH + BC_EAST) / 2;
    heat[(ny - 1) * nx + nx - 1] = (BC_NORTH + BC_EAST) / 2;
    for (size_t y = 1; y <= my; y++)
        for (size_t xx = 1; xx <= mx; xx++) heat[y * nx + xx] = x[(y - 1) * mx + (xx - 1)];
    File fh(hpath, "wb");
    if (!fh.f || !write_header(fh, ny, nx) || fwrite(heat.data(), sizeof(double), nx * ny, fh) != nx * ny || !fh.close()) {
        fprintf(stderr, "Cannot write heat file\n");
        return 2;
    }
    return 0;
}
}  // namespace

int main(int argc, char **argv)
{
    if (argc == 6 && !strcmp(argv[1], "assemble")) return assemble(atoll(argv[2]), atoll(argv[3]), argv[4], argv[5]);
    if (argc == 6 && !strcmp(argv[1], "field")) return field(atoll(argv[2]), atoll(argv[3]), argv[4], argv[5]);
    fprintf(stderr, "usage: %s assemble nx ny matrix.bin rhs.bin | field nx ny sol.bin heat.bin\n", argv[0]);
    return 1;
}
