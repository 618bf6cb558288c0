// The ranks of one multi-GPU run of the CLI: one process per GPU, forked from the process
// the user started before anything touches a GPU, connected by socket pairs.
//
// The reference's main (diffusion_stereo/src/main.rs:36-98) is one process; here rank r owns
// the r-th row band of the image.  The sockets carry what the host needs to share -- the
// RCCL communicator's id, each band's labels on their way to rank 0 -- and, for ranks that
// share one GPU (rehearsal on a one-GPU machine) or when RCCL is unavailable, the halo rows
// themselves (host_transport): the library's band protocol takes any ds_transport.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "diffstereo.h"

namespace stereo {

class RankGroup {
public:
    // Forks world - 1 children.  Returns in every process with its rank set (the caller's
    // process is rank 0).  Must run before the first HIP call of the process.
    static RankGroup fork_ranks(int world);
    ~RankGroup();
    RankGroup(RankGroup &&o) noexcept;
    RankGroup &operator=(RankGroup &&o) noexcept;
    RankGroup(const RankGroup &) = delete;

    int rank() const { return rank_; }
    int world() const { return world_; }
    int up() const { return rank_ > 0 ? rank_ - 1 : -1; }
    int down() const { return rank_ + 1 < world_ ? rank_ + 1 : -1; }

    void broadcast(void *data, std::size_t bytes);                      // from rank 0 to all
    // rank 0 receives every rank's block (rank order, its own first); sizes may differ
    std::vector<std::vector<std::uint8_t>> gather(const void *data, std::size_t bytes);
    // Collective: true on every rank iff every rank passed true.  A rank that dies instead of
    // calling it counts as false (its socket reads as closed), so the others leave together
    // before they enter a collective the dead rank would never join.
    bool agree(bool ok);
    // rank 0: waits for the children in the order they exit; as soon as one ends badly the
    // others get SIGTERM (they may be waiting in a collective for the one that is gone).
    // Returns 0 if all exited with status 0.
    int wait_children();
    // rank 0: SIGTERM to every child still running (rank 0 itself has failed)
    void terminate_children();

    // Halo rows and reductions through these sockets and host memory (blocking).
    ds_transport host_transport();
    std::string error;

private:
    RankGroup() = default;
    int rank_ = 0, world_ = 1;
    int up_fd_ = -1, down_fd_ = -1;      // neighbours in the chain of bands
    int hub_fd_ = -1;                    // ranks > 0: to rank 0
    std::vector<int> spoke_fd_;          // rank 0: to rank r at [r]
    std::vector<int> children_;
    std::vector<std::uint8_t> stage_out_, stage_in_;

    static int exchange_cb(void *ctx, int n, const int *peers, void *const *send, void *const *recv,
                           const std::size_t *bytes, void *hip_stream);
    static int all_reduce_cb(void *ctx, void *dev, int count, int kind, void *hip_stream);
    static const char *last_error_cb(void *ctx);
    int fd_of(int peer) const { return peer == rank_ - 1 ? up_fd_ : peer == rank_ + 1 ? down_fd_ : -1; }
};

}  // namespace stereo
