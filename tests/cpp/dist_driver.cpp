// One rank of the multi-GPU result gather through the C++ adaptor (snake_hip::Dist, RCCL over xGMI behind the C ABI: snk_dist_*).
//   dist_driver <rendezvous file> <rank> <world> <device> <out file>
// Every rank makes a TUM trajectory that depends on its rank (rank-dependent length: the padding is exercised), gathers all of them
// and writes what it received; the test (tests/test_dist_gpu.py) checks every rank's file against the expected rows.  Plain g++: no
// HIP in this translation unit.
#include <cstdio>
#include <cstdlib>

#include "snake_hip.hpp"

static std::vector<snake_hip::TumPose> trajectory_of(int rank)
{
    // what System::writeFrameTrajectory would write (Snake/System/System.cpp:546-563): timestamp, translation, unit quaternion
    std::vector<snake_hip::TumPose> tr((size_t)(37 + 11 * rank));
    for (size_t i = 0; i < tr.size(); ++i)
    {
        const double a = 0.01 * (double)i + (double)rank;
        tr[i]          = {1403636579.0 + 0.05 * (double)i, a, 2.0 * a, -a, 0.0, 0.0, std::sin(a / 2), std::cos(a / 2)};
    }
    return tr;
}

int main(int argc, char** argv)
{
    if (argc != 6) return 2;
    const int rank = atoi(argv[2]), world = atoi(argv[3]), device = atoi(argv[4]);
    try
    {
        int version = 0;
        snake_hip::check(snk_dist_rccl_version(&version), "snk_dist_rccl_version");
        snake_hip::Dist dist(argv[1], rank, world, device, 120.0);
        const auto all = dist.GatherTrajectories(trajectory_of(rank));
        struct Stats { double frames, frames_per_s, matches, cost; };
        const auto stats = dist.GatherBlocks(Stats{(double)(37 + 11 * rank), 1000.0 + rank, 250.0 * rank, 0.5});
        FILE* f = fopen(argv[5], "w");
        if (!f) return 3;
        int c_rank = -1, c_world = -1;
        snake_hip::check(snk_dist_rank(dist.handle(), &c_rank, &c_world), "snk_dist_rank");  // ncclCommUserRank / ncclCommCount of the communicator
        if (c_rank != rank || c_world != world) return 4;
        fprintf(f, "rccl %d world %d\n", version, c_world);
        for (int r = 0; r < world; ++r)
        {
            const auto want = trajectory_of(r);
            bool same       = all[(size_t)r].size() == want.size();
            for (size_t i = 0; same && i < want.size(); ++i) same = std::memcmp(&all[(size_t)r][i], &want[i], sizeof(want[i])) == 0;
            fprintf(f, "rank %d rows %zu %s stats %.0f %.0f %.0f\n", r, all[(size_t)r].size(), same ? "identical" : "DIFFERENT",
                    stats[(size_t)r].frames, stats[(size_t)r].frames_per_s, stats[(size_t)r].matches);
            // the rows themselves, the way the reference prints them (precision 15)
            if (r == rank)
                for (const auto& p : all[(size_t)r]) fprintf(f, "%.15g %.15g %.15g %.15g %.15g %.15g %.15g %.15g\n", p.t, p.tx, p.ty, p.tz, p.qx, p.qy, p.qz, p.qw);
        }
        fclose(f);
    }
    catch (const std::exception& e)
    {
        fprintf(stderr, "dist_driver rank %d: %s\n", rank, e.what());
        return 1;
    }
    return 0;
}
