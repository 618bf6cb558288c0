// stereovision: left/right ASCII PGM in, disparity PGM out -- the CLI contract of the
// reference's main (diffusion_stereo/src/main.rs:36-98), with its hard-coded constants
// as defaults: max disparity 40, smoothing 1.2, epsilon 1/(10*W*H), batches of 200.
//
//   stereovision LEFT.pgm RIGHT.pgm OUT.pgm [--max-disparity N] [--smoothing X]
//                [--epsilon E] [--batch B] [--max-batches M] [--dtype f64|f32]
//                [--decode crossing|crossing-host|belief|epsilon-search|epsilon-search-streamed] [--iterations K]
//                [--tolerance T]
//                [--schedule red-black|raster] [--labeling simple|best] [--quiet]
//                [--gpus N] [--transport rccl|host] [--share-gpu]
//
// --gpus N splits the image into N row bands, one process per GPU (forked before anything
// touches a GPU); the library runs the band protocol (ds_band_iterate, ds_band_energy,
// ds_band_crossing_out) with its halo rows over RCCL, rank 0 gathers the labels and writes the
// PGM.  --transport host moves the halo rows through host memory and sockets instead;
// --share-gpu puts all ranks on device 0 (a rehearsal on a one-GPU machine; needs --transport
// host, RCCL refuses two ranks on one device).  Bands decode with --decode crossing (default)
// and --labeling simple, as the reference's main does.
//
// --decode crossing (default) is the reference pipeline: diffusion batches on the GPU
// until the epsilon-consistent sub-graph is non-empty (crossing out on the GPU too), then
// the smallest surviving label per pixel.  --decode crossing-host keeps the reference's
// vertex and edge masks on the host instead (O(H*W*D^2) bytes: small images only).
// --decode belief runs --iterations diffusion iterations and takes the per-pixel argmin.
// --decode epsilon-search (epsilon_search.rs, disabled in the reference's main) runs
// --iterations iterations, collects all penalty gaps (small images only; merged within
// --tolerance), binary-searches the smallest epsilon with a non-empty consistent sub-graph
// and labels with it.  --decode epsilon-search-streamed is the same search with tolerance 0
// for images of any size: no array, the gaps are walked on the GPU once per question.
// --labeling best takes find_best_labeling (the alternative main.rs:64 keeps commented out)
// instead of simple_best_labeling: one label per pixel fixed in raster order, with a
// crossing-out after each choice.
// --schedule raster sweeps in the reference's sequential order (same results as its
// CPU implementation up to rounding; one small launch per anti-diagonal, so slower).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <optional>
#include <string>

#include <unistd.h>

#include "crossing_out_graph.hpp"
#include "diffusion_graph.hpp"
#include "epsilon_search.hpp"
#include "pgm.hpp"
#include "rank_group.hpp"

namespace {

// main over `world` row bands; every rank runs it, rank 0 returns the whole map.
stereo::Matrix run_bands(stereo::RankGroup &group, const stereo::PgmImage &left, const stereo::PgmImage &right,
                         std::size_t max_disparity, double smoothing, int dtype, double epsilon, std::size_t batch,
                         std::size_t max_batches, bool verbose, bool rccl, bool share_gpu)
{
    using namespace stereo;
    const int rank = group.rank(), world = group.world();
    const std::size_t H = left.height;
    if ((std::size_t)world > H) throw std::runtime_error("more bands than image rows");
    const std::size_t base = H / world, extra = H % world;
    const std::size_t rb = rank * base + std::min<std::size_t>(rank, extra), re = rb + base + ((std::size_t)rank < extra ? 1 : 0);
    const int device = share_gpu ? 0 : rank;
    // Every step that one rank alone can fail at (no such device, out of memory) is followed by
    // an agreement over the sockets: a rank must not enter a collective -- the communicator's
    // creation, an exchange -- that a failed rank will never join.
    std::string failure;
    std::optional<DiffusionGraph> made_graph;
    try {
        made_graph.emplace(DiffusionGraph::initialize_band(left.matrix, right.matrix, max_disparity, smoothing, dtype, rb,
                                                           re, device));
    } catch (const std::exception &e) {
        failure = e.what();
    }
    if (!group.agree(made_graph.has_value()))
        throw std::runtime_error(made_graph ? "another rank could not set up its band" : failure);
    DiffusionGraph graph = std::move(*made_graph);
    ds_comm *comm = nullptr;
    ds_transport host{};
    if (rccl) {
        std::uint8_t id[DS_COMM_ID_BYTES] = {0};
        int ok = rank != 0 || ds_comm_unique_id(id) == DS_OK;
        group.broadcast(&ok, sizeof ok);
        if (!ok) throw std::runtime_error(std::string("RCCL: ") + ds_comm_last_error(nullptr));
        group.broadcast(id, sizeof id);
        const bool made = ds_comm_create(id, rank, world, device, &comm) == DS_OK;
        if (!made) failure = std::string("RCCL: ") + ds_comm_last_error(nullptr);
        if (!group.agree(made)) {
            ds_comm_destroy(comm);
            throw std::runtime_error(made ? "another rank could not join the communicator" : failure);
        }
        graph.attach(comm, group.up(), group.down());
    } else {
        host = group.host_transport();
        graph.attach(host, group.up(), group.down());
    }
    Matrix whole;
    {
        CrossingOutGraph cog = CrossingOutGraph::initialize(std::move(graph), true);
        // the same `verbose` on every rank: the energies it prints are collectives
        cog.diffusion_while_not_consistent(epsilon, batch, max_batches, verbose);
        const Matrix mine = cog.simple_best_labeling();
        std::vector<std::uint32_t> flat;
        for (const auto &row : mine)
            for (std::size_t v : row) flat.push_back((std::uint32_t)v);
        const auto parts = group.gather(flat.data(), flat.size() * sizeof(std::uint32_t));
        if (rank == 0) {
            for (const auto &part : parts) {
                const std::uint32_t *v = reinterpret_cast<const std::uint32_t *>(part.data());
                const std::size_t rows = part.size() / sizeof(std::uint32_t) / left.width;
                for (std::size_t i = 0; i < rows; ++i) whole.emplace_back(v + i * left.width, v + (i + 1) * left.width);
            }
            if (whole.size() != H) throw std::runtime_error("gathered labels do not cover the image");
        }
    }  // the handle goes before the communicator it uses
    ds_comm_destroy(comm);
    return whole;
}

}  // namespace

int main(int argc, char **argv)
{
    using namespace stereo;
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s LEFT.pgm RIGHT.pgm OUT.pgm [options]\n", argv[0]);
        return 2;
    }
    const std::string left_path = argv[1], right_path = argv[2], out_path = argv[3];
    std::size_t max_disparity = 40, batch = 200, max_batches = 1000, iterations = 200;
    double smoothing = 1.2, epsilon = -1., tolerance = 1e-9;
    int dtype = DS_F64;
    bool crossing = true, host_masks = false, verbose = true, raster = false, search = false, streamed = false, best = false;
    int gpus = 1;
    bool rccl = true, share_gpu = false;
    for (int a = 4; a < argc; ++a) {
        const std::string o = argv[a];
        auto need = [&](const char *name) -> const char * {
            if (a + 1 >= argc) {
                std::fprintf(stderr, "%s needs a value\n", name);
                std::exit(2);
            }
            return argv[++a];
        };
        if (o == "--max-disparity") max_disparity = std::strtoul(need("--max-disparity"), nullptr, 10);
        else if (o == "--smoothing") smoothing = std::strtod(need("--smoothing"), nullptr);
        else if (o == "--epsilon") epsilon = std::strtod(need("--epsilon"), nullptr);
        else if (o == "--batch") batch = std::strtoul(need("--batch"), nullptr, 10);
        else if (o == "--max-batches") max_batches = std::strtoul(need("--max-batches"), nullptr, 10);
        else if (o == "--iterations") iterations = std::strtoul(need("--iterations"), nullptr, 10);
        else if (o == "--dtype") dtype = std::strcmp(need("--dtype"), "f32") == 0 ? DS_F32 : DS_F64;
        else if (o == "--decode") {
            const std::string mode = need("--decode");
            if (mode != "crossing" && mode != "crossing-host" && mode != "belief" && mode != "epsilon-search" &&
                mode != "epsilon-search-streamed") {
                std::fprintf(stderr, "unknown --decode %s\n", mode.c_str());
                return 2;
            }
            crossing = mode != "belief";
            host_masks = mode == "crossing-host";
            search = mode == "epsilon-search" || mode == "epsilon-search-streamed";
            streamed = mode == "epsilon-search-streamed";
        }
        else if (o == "--tolerance") tolerance = std::strtod(need("--tolerance"), nullptr);
        else if (o == "--schedule") {
            const std::string mode = need("--schedule");
            if (mode != "raster" && mode != "red-black") {
                std::fprintf(stderr, "unknown --schedule %s\n", mode.c_str());
                return 2;
            }
            raster = mode == "raster";
        }
        else if (o == "--labeling") {
            const std::string mode = need("--labeling");
            if (mode != "simple" && mode != "best") {
                std::fprintf(stderr, "unknown --labeling %s\n", mode.c_str());
                return 2;
            }
            best = mode == "best";
        }
        else if (o == "--quiet") verbose = false;
        else if (o == "--gpus") gpus = std::atoi(need("--gpus"));
        else if (o == "--share-gpu") share_gpu = true;
        else if (o == "--transport") {
            const std::string mode = need("--transport");
            if (mode != "rccl" && mode != "host") {
                std::fprintf(stderr, "unknown --transport %s\n", mode.c_str());
                return 2;
            }
            rccl = mode == "rccl";
        }
        else {
            std::fprintf(stderr, "unknown option %s\n", o.c_str());
            return 2;
        }
    }
    if (gpus < 1 || (gpus > 1 && (!crossing || host_masks || search || best || raster)) || (share_gpu && rccl && gpus > 1)) {
        std::fprintf(stderr, "--gpus N needs --decode crossing, --labeling simple, --schedule red-black; "
                             "--share-gpu needs --transport host\n");
        return 2;
    }
    if (gpus > 1) {
        int status = 1;
        try {
            RankGroup group = RankGroup::fork_ranks(gpus);  // before the first HIP call
            // every rank runs the same code, prints included; only rank 0's reach the terminal
            if (group.rank() != 0 && !std::freopen("/dev/null", "w", stdout)) _exit(1);
            try {
                const PgmImage right = pgm_reader(right_path);
                const PgmImage left = pgm_reader(left_path);
                if (right.width != left.width || right.height != left.height)
                    throw std::runtime_error("left and right images differ in size");
                if (epsilon < 0.) epsilon = 1. / double(10 * left.width * left.height);
                if (group.rank() == 0) std::printf("Epsilon: %g\n", epsilon);
                const Matrix map = run_bands(group, left, right, max_disparity, smoothing, dtype, epsilon, batch,
                                             max_batches, verbose, rccl, share_gpu);
                if (group.rank() == 0) {
                    std::printf("Disparity map is consistent: %s\n", check_disparity_map(map) ? "true" : "false");
                    pgm_writer(map, out_path, max_disparity);
                    std::printf("Disparity map is saved to `%s`\n", out_path.c_str());
                }
                status = 0;  // rank 0: only once the map is on disk
            } catch (const std::exception &e) {
                std::fprintf(stderr, "error (rank %d): %s\n", group.rank(), e.what());
                if (group.rank() == 0) group.terminate_children();  // they may wait for this rank in a collective
            }
            std::fflush(stdout);
            if (group.rank() != 0) _exit(status);  // children: no atexit handlers of the parent's state
            status |= group.wait_children();
        } catch (const std::exception &e) {
            std::fprintf(stderr, "error: %s\n", e.what());
        }
        return status;
    }
    try {
        const PgmImage right = pgm_reader(right_path);
        const PgmImage left = pgm_reader(left_path);
        if (right.width != left.width || right.height != left.height)
            throw std::runtime_error("left and right images differ in size");
        if (epsilon < 0.) epsilon = 1. / double(10 * left.width * left.height);
        std::printf("Epsilon: %g\n", epsilon);
        DiffusionGraph graph = DiffusionGraph::initialize(left.matrix, right.matrix, max_disparity, smoothing, dtype);
        graph.raster_order = raster;
        Matrix disparity_map;
        if (search) {
            CrossingOutGraph cog = CrossingOutGraph::initialize(std::move(graph), true);
            cog.diffusion_graph.diffusion(1, iterations, verbose);
            double found = 0.;
            if (streamed) {
                found = epsilon_search_streamed(cog);
            } else {
                const std::vector<double> array = create_array_of_epsilons(cog, tolerance);
                std::printf("Array size: %zu\n", array.size());
                found = epsilon_search(cog, array);
            }
            std::printf("Smallest consistent epsilon: %.17g\n", found);
            if (!cog.consistent_with_epsilon(found)) throw std::runtime_error("no epsilon in the array is consistent");
            disparity_map = best ? cog.find_best_labeling() : cog.simple_best_labeling();
        } else if (crossing) {
            CrossingOutGraph cog = CrossingOutGraph::initialize(std::move(graph), !host_masks);
            cog.diffusion_while_not_consistent(epsilon, batch, max_batches, verbose);
            std::printf("Finding disparity map ...\n");
            disparity_map = best ? cog.find_best_labeling() : cog.simple_best_labeling();
        } else {
            graph.diffusion(1, iterations, verbose);
            std::printf("Finding disparity map ...\n");
            disparity_map = graph.belief_depth_map();
        }
        std::printf("Disparity map is consistent: %s\n", check_disparity_map(disparity_map) ? "true" : "false");
        pgm_writer(disparity_map, out_path, max_disparity);
        std::printf("Disparity map is saved to `%s`\n", out_path.c_str());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
