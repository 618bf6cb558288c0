#include "rank_group.hpp"

#include <hip/hip_runtime_api.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>
#include <csignal>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <utility>

namespace stereo {

namespace {

void write_all(int fd, const void *p, std::size_t n)
{
    const char *c = static_cast<const char *>(p);
    while (n) {
        const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
        if (k < 0) {
            if (errno == EINTR) continue;
            throw std::runtime_error(std::string("send to a peer rank failed: ") + std::strerror(errno));
        }
        c += k;
        n -= (std::size_t)k;
    }
}

void read_all(int fd, void *p, std::size_t n)
{
    char *c = static_cast<char *>(p);
    while (n) {
        const ssize_t k = ::recv(fd, c, n, 0);
        if (k == 0) throw std::runtime_error("a peer rank closed its socket (it has failed)");
        if (k < 0) {
            if (errno == EINTR) continue;
            throw std::runtime_error(std::string("recv from a peer rank failed: ") + std::strerror(errno));
        }
        c += k;
        n -= (std::size_t)k;
    }
}

}  // namespace

RankGroup RankGroup::fork_ranks(int world)
{
    if (world < 1) throw std::invalid_argument("need at least one rank");
    RankGroup g;
    g.world_ = world;
    if (world == 1) return g;
    // chain[r] joins ranks r and r + 1; hub[r] joins ranks 0 and r
    std::vector<std::pair<int, int>> chain(world - 1), hub(world);
    for (auto &p : chain) {
        int fds[2];
        if (socketpair(AF_UNIX, SOCK_STREAM, 0, fds) != 0) throw std::runtime_error("socketpair failed");
        p = {fds[0], fds[1]};
    }
    for (int r = 1; r < world; ++r) {
        int fds[2];
        if (socketpair(AF_UNIX, SOCK_STREAM, 0, fds) != 0) throw std::runtime_error("socketpair failed");
        hub[r] = {fds[0], fds[1]};
    }
    int me = 0;
    for (int r = 1; r < world; ++r) {
        const pid_t pid = fork();
        if (pid < 0) throw std::runtime_error("fork failed");
        if (pid == 0) {
            me = r;
            g.children_.clear();
            break;
        }
        g.children_.push_back((int)pid);
    }
    g.rank_ = me;
    for (int r = 0; r + 1 < world; ++r) {  // keep my ends, close the others
        if (r == me) g.down_fd_ = chain[r].first; else close(chain[r].first);
        if (r + 1 == me) g.up_fd_ = chain[r].second; else close(chain[r].second);
    }
    g.spoke_fd_.assign(world, -1);
    for (int r = 1; r < world; ++r) {
        if (me == 0) g.spoke_fd_[r] = hub[r].first; else close(hub[r].first);
        if (me == r) g.hub_fd_ = hub[r].second; else close(hub[r].second);
    }
    return g;
}

RankGroup::RankGroup(RankGroup &&o) noexcept { *this = std::move(o); }

RankGroup &RankGroup::operator=(RankGroup &&o) noexcept
{
    rank_ = o.rank_; world_ = o.world_;
    up_fd_ = o.up_fd_; down_fd_ = o.down_fd_; hub_fd_ = o.hub_fd_;
    spoke_fd_ = std::move(o.spoke_fd_);
    children_ = std::move(o.children_);
    o.up_fd_ = o.down_fd_ = o.hub_fd_ = -1;
    o.spoke_fd_.clear();
    o.children_.clear();
    return *this;
}

RankGroup::~RankGroup()
{
    for (int fd : {up_fd_, down_fd_, hub_fd_})
        if (fd >= 0) close(fd);
    for (int fd : spoke_fd_)
        if (fd >= 0) close(fd);
}

void RankGroup::broadcast(void *data, std::size_t bytes)
{
    if (world_ == 1) return;
    if (rank_ == 0)
        for (int r = 1; r < world_; ++r) write_all(spoke_fd_[r], data, bytes);
    else
        read_all(hub_fd_, data, bytes);
}

std::vector<std::vector<std::uint8_t>> RankGroup::gather(const void *data, std::size_t bytes)
{
    std::vector<std::vector<std::uint8_t>> out;
    const std::uint8_t *p = static_cast<const std::uint8_t *>(data);
    if (rank_ != 0) {
        const std::uint64_t n = bytes;
        write_all(hub_fd_, &n, sizeof n);
        write_all(hub_fd_, data, bytes);
        return out;
    }
    out.emplace_back(p, p + bytes);
    for (int r = 1; r < world_; ++r) {
        std::uint64_t n = 0;
        read_all(spoke_fd_[r], &n, sizeof n);
        out.emplace_back((std::size_t)n);
        read_all(spoke_fd_[r], out.back().data(), (std::size_t)n);
    }
    return out;
}

bool RankGroup::agree(bool ok)
{
    if (world_ == 1) return ok;
    std::uint8_t all = ok ? 1 : 0;
    if (rank_ == 0) {
        for (int r = 1; r < world_; ++r) {
            std::uint8_t theirs = 0;
            try {
                read_all(spoke_fd_[r], &theirs, 1);
            } catch (const std::exception &) {
                theirs = 0;  // the rank is gone
            }
            all = all && theirs;
        }
        for (int r = 1; r < world_; ++r) {
            try {
                write_all(spoke_fd_[r], &all, 1);
            } catch (const std::exception &) {
            }
        }
    } else {
        try {
            write_all(hub_fd_, &all, 1);
            read_all(hub_fd_, &all, 1);
        } catch (const std::exception &) {
            all = 0;  // rank 0 is gone
        }
    }
    return all != 0;
}

void RankGroup::terminate_children()
{
    for (int pid : children_) (void)kill((pid_t)pid, SIGTERM);
}

int RankGroup::wait_children()
{
    int bad = 0;
    std::size_t left = children_.size();
    while (left > 0) {
        int status = 0;
        const pid_t pid = waitpid(-1, &status, 0);
        if (pid < 0) {
            if (errno == EINTR) continue;
            bad = 1;  // no child left to wait for although some were expected
            break;
        }
        bool mine = false;
        for (int &c : children_)
            if (c == (int)pid) {
                c = -1;
                mine = true;
            }
        if (!mine) continue;
        --left;
        if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) {
            if (!bad)
                for (int c : children_)
                    if (c > 0) (void)kill((pid_t)c, SIGTERM);
            bad = 1;
        }
    }
    children_.clear();
    return bad;
}

// Every pair of neighbours handles its transfers in the same order, the lower rank sending
// first and the higher rank receiving first, and every rank serves its upper neighbour before
// its lower one: no two ranks ever wait for each other.
int RankGroup::exchange_cb(void *ctx, int n, const int *peers, void *const *send, void *const *recv,
                           const std::size_t *bytes, void *hip_stream)
{
    RankGroup *g = static_cast<RankGroup *>(ctx);
    try {
        if (hipStreamSynchronize(static_cast<hipStream_t>(hip_stream)) != hipSuccess)
            throw std::runtime_error("hipStreamSynchronize failed");
        for (int pass = 0; pass < 2; ++pass) {  // 0: the neighbour above, 1: the neighbour below
            const int peer = pass == 0 ? g->up() : g->down();
            if (peer < 0) continue;
            const int fd = g->fd_of(peer);
            for (int k = 0; k < n; ++k) {
                if (peers[k] != peer) continue;
                if (g->fd_of(peers[k]) < 0) throw std::runtime_error("transfer to a rank that is no neighbour");
                g->stage_out_.resize(bytes[k]);
                g->stage_in_.resize(bytes[k]);
                if (hipMemcpy(g->stage_out_.data(), send[k], bytes[k], hipMemcpyDeviceToHost) != hipSuccess)
                    throw std::runtime_error("hipMemcpy (device to host) failed");
                if (g->rank_ < peer) {
                    write_all(fd, g->stage_out_.data(), bytes[k]);
                    read_all(fd, g->stage_in_.data(), bytes[k]);
                } else {
                    read_all(fd, g->stage_in_.data(), bytes[k]);
                    write_all(fd, g->stage_out_.data(), bytes[k]);
                }
                if (hipMemcpy(recv[k], g->stage_in_.data(), bytes[k], hipMemcpyHostToDevice) != hipSuccess)
                    throw std::runtime_error("hipMemcpy (host to device) failed");
            }
        }
        for (int k = 0; k < n; ++k)
            if (peers[k] != g->up() && peers[k] != g->down()) throw std::runtime_error("transfer to a rank that is no neighbour");
        return 0;
    } catch (const std::exception &e) {
        g->error = e.what();
        return 1;
    }
}

int RankGroup::all_reduce_cb(void *ctx, void *dev, int count, int kind, void *hip_stream)
{
    RankGroup *g = static_cast<RankGroup *>(ctx);
    try {
        const bool f64 = kind == DS_REDUCE_SUM_F64 || kind == DS_REDUCE_MAX_F64;
        if (!f64 && kind != DS_REDUCE_MAX_I32 && kind != DS_REDUCE_MIN_I32) throw std::runtime_error("unknown reduction kind");
        const std::size_t bytes = (std::size_t)count * (f64 ? sizeof(double) : sizeof(int));
        std::vector<std::uint8_t> mine(bytes), other(bytes);
        if (hipStreamSynchronize(static_cast<hipStream_t>(hip_stream)) != hipSuccess ||
            hipMemcpy(mine.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess)
            throw std::runtime_error("reading the values to reduce failed");
        auto fold = [&](const std::uint8_t *src) {
            for (int k = 0; k < count; ++k) {
                if (f64) {
                    double a, b;
                    std::memcpy(&a, mine.data() + k * sizeof a, sizeof a);
                    std::memcpy(&b, src + k * sizeof b, sizeof b);
                    a = kind == DS_REDUCE_SUM_F64 ? a + b : (a > b ? a : b);
                    std::memcpy(mine.data() + k * sizeof a, &a, sizeof a);
                } else {
                    int a, b;
                    std::memcpy(&a, mine.data() + k * sizeof a, sizeof a);
                    std::memcpy(&b, src + k * sizeof b, sizeof b);
                    a = kind == DS_REDUCE_MAX_I32 ? (a > b ? a : b) : (a < b ? a : b);
                    std::memcpy(mine.data() + k * sizeof a, &a, sizeof a);
                }
            }
        };
        if (g->rank_ == 0) {  // rank order: the sum is the same whatever the timing
            for (int r = 1; r < g->world_; ++r) {
                read_all(g->spoke_fd_[r], other.data(), bytes);
                fold(other.data());
            }
            for (int r = 1; r < g->world_; ++r) write_all(g->spoke_fd_[r], mine.data(), bytes);
        } else {
            write_all(g->hub_fd_, mine.data(), bytes);
            read_all(g->hub_fd_, mine.data(), bytes);
        }
        if (hipMemcpy(dev, mine.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
            throw std::runtime_error("writing the reduced values failed");
        return 0;
    } catch (const std::exception &e) {
        g->error = e.what();
        return 1;
    }
}

const char *RankGroup::last_error_cb(void *ctx) { return static_cast<RankGroup *>(ctx)->error.c_str(); }

ds_transport RankGroup::host_transport()
{
    ds_transport t;
    t.ctx = this;
    t.exchange = exchange_cb;
    t.all_reduce = all_reduce_cb;
    t.last_error = last_error_cb;
    return t;
}

}  // namespace stereo
