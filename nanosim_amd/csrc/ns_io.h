// ns_io.h — the output side of a worker (host code only): result buffers leave HBM for files WHILE the next batch is generated.
//
// The reference's workers write every record and every error-profile row as they go (out_reads.write S:1437-1443, out_error.write
// S:2006-2008).  Here a batch's FASTA/FASTQ image and error-profile image are complete in HBM when ns_generate returns; ns_sink_write
// queues them for a file and returns at once:
//   copier thread   cuts the buffer into slices, takes a free page-locked staging slice, issues hipMemcpyAsync on the context's COPY
//                   stream (its own HIP stream: DMA engines, no kernel) and records an event behind it; several slices are in flight
//   writer threads  wait for a slice's event, pwrite() the bytes at their final file offset, hand the staging slice back; ONE writer per
//                   file at a time (writes to one inode serialise in the kernel: measured on tmpfs, 8 threads on one file reach
//                   3.4 GB/s where one thread reaches 6 GB/s, while 16 files take 70 GB/s) — parallelism comes from several sinks
// The context keeps TWO result slots (record + error-profile buffers): while slot s is being copied out, the next ns_generate fills
// slot s ^ 1; the one after that waits until the copies out of s have left the device (not until they are in the file).
#pragma once
#include <hip/hip_runtime.h>
#include <errno.h>
#include <string.h>
#include <unistd.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

struct ns_sink {
    int fd = -1;                          // -1: the bytes are copied to the host and dropped (measures the device-to-host rate alone)
    uint64_t off = 0;                     // file offset of the next byte queued
    std::atomic<uint64_t> queued{0}, written{0};
    std::atomic<int> err{0};              // errno of the first failed write
    bool writing = false;                 // a writer thread is inside pwrite() for this sink (guarded by the engine's mutex)
};

struct IoEngine {
    int device = 0;
    hipStream_t stream = nullptr;
    size_t slice_bytes = 0;
    struct Slice { uint8_t *pin = nullptr; hipEvent_t t0 = nullptr, t1 = nullptr; };
    std::vector<Slice> slices;
    struct CopyJob { const uint8_t *src; uint64_t n; ns_sink *sink; uint64_t file_off; int slot; };
    struct WriteJob { int slice; ns_sink *sink; uint64_t file_off; size_t n; int slot; };
    std::mutex mu;
    std::condition_variable cv_free, cv_copy, cv_write, cv_idle;
    std::deque<int> free_slices;
    std::deque<CopyJob> copy_jobs;
    std::deque<WriteJob> copied_jobs;      // slices whose copy has been issued, oldest first
    std::deque<WriteJob> ready_jobs;       // slices that have arrived in their staging memory
    uint64_t slot_pending[2] = {0, 0};    // slices of the slot that have not left the device yet
    uint64_t jobs_open = 0;               // slices queued and not yet written
    bool stop = false;
    std::string err;                      // first HIP error of the copier / writers
    std::thread copier;
    std::vector<std::thread> writers;
    // accounting (ns_io_counters): bytes and DMA time of the slices, time the copier waited for a free staging slice
    double dma_ms = 0, wait_free_s = 0, write_s = 0;
    uint64_t bytes = 0;

    static size_t env_or(const char *name, size_t dflt) { const char *v = getenv(name); return v && atoll(v) > 0 ? (size_t)atoll(v) : dflt; }

    int start(int dev, std::string &msg) {
        device = dev;
        slice_bytes = env_or("NS_IO_SLICE_BYTES", env_or("NS_IO_SLICE_MB", 16) << 20);      // (bytes: tests that want many slices per small batch)
        const size_t n_slices = env_or("NS_IO_SLICES", 24), n_threads = env_or("NS_IO_THREADS", 16);
        if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) { msg = "ns_io: no copy stream"; return -1; }
        slices.resize(n_slices);
        for (size_t i = 0; i < n_slices; ++i) {
            Slice &s = slices[i];
            if (hipHostMalloc((void **)&s.pin, slice_bytes, hipHostMallocDefault) != hipSuccess || hipEventCreate(&s.t0) != hipSuccess ||
                hipEventCreate(&s.t1) != hipSuccess) { msg = "ns_io: staging allocation failed"; return -1; }
            free_slices.push_back((int)i);
        }
        copier = std::thread([this] { copy_loop(); });
        for (size_t t = 0; t < n_threads; ++t) writers.emplace_back([this] { write_loop(); });
        return 0;
    }
    void shutdown() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv_copy.notify_all(); cv_write.notify_all(); cv_free.notify_all();
        if (copier.joinable()) copier.join();
        for (auto &t : writers) if (t.joinable()) t.join();
        hipError_t e = hipSetDevice(device); (void)e;
        if (stream) { e = hipStreamSynchronize(stream); e = hipStreamDestroy(stream); }
        for (Slice &s : slices) {
            if (s.pin) e = hipHostFree(s.pin);
            if (s.t0) e = hipEventDestroy(s.t0);
            if (s.t1) e = hipEventDestroy(s.t1);
        }
        slices.clear();
    }
    void fail(const std::string &m) { std::lock_guard<std::mutex> g(mu); if (err.empty()) err = m; }

    // queue bytes [0, n) of a device buffer of result slot `slot` for sink `s`
    void enqueue(ns_sink *s, const void *src, uint64_t n, int slot) {
        if (!n) return;
        const uint64_t n_sl = (n + slice_bytes - 1) / slice_bytes;
        {
            std::lock_guard<std::mutex> g(mu);
            copy_jobs.push_back(CopyJob{static_cast<const uint8_t *>(src), n, s, s->off, slot});
            slot_pending[slot] += n_sl; jobs_open += n_sl;
        }
        s->off += n; s->queued += n;
        cv_copy.notify_one();
    }
    void wait_slot(int slot) { std::unique_lock<std::mutex> g(mu); cv_idle.wait(g, [&] { return slot_pending[slot] == 0; }); }
    bool slot_busy(int slot) { std::lock_guard<std::mutex> g(mu); return slot_pending[slot] != 0; }
    void wait_all() { std::unique_lock<std::mutex> g(mu); cv_idle.wait(g, [&] { return jobs_open == 0; }); }
    void wait_sink(ns_sink *s) { std::unique_lock<std::mutex> g(mu); cv_idle.wait(g, [&] { return s->written.load() == s->queued.load(); }); }

    // The copier takes the slices of the queued buffers ROUND-ROBIN over the buffers (up to NS_IO_FANOUT of them): a batch cut into K
    // sub-files has slices of K files in flight, so K writers are busy — taken one buffer after the other, all staging slices would
    // belong to one file at a time and its single writer would set the pace.
    void copy_loop() {
        hipError_t e = hipSetDevice(device); (void)e;
        const size_t fanout = env_or("NS_IO_FANOUT", 32);
        struct Active { CopyJob j; uint64_t pos; };
        std::vector<Active> act;
        size_t rr = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(mu);
                if (act.empty()) cv_copy.wait(g, [&] { return stop || !copy_jobs.empty(); });
                while (act.size() < fanout && !copy_jobs.empty()) { act.push_back(Active{copy_jobs.front(), 0}); copy_jobs.pop_front(); }
                if (act.empty()) return;
            }
            if (rr >= act.size()) rr = 0;
            Active &a = act[rr];
            const size_t n = (size_t)std::min<uint64_t>(slice_bytes, a.j.n - a.pos);
            int si;
            {
                std::unique_lock<std::mutex> g(mu);
                const auto t0 = std::chrono::steady_clock::now();
                cv_free.wait(g, [&] { return stop || !free_slices.empty(); });
                wait_free_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (free_slices.empty()) return;
                si = free_slices.front(); free_slices.pop_front();
            }
            Slice &s = slices[si];
            hipError_t e1 = hipEventRecord(s.t0, stream);
            hipError_t e2 = hipMemcpyAsync(s.pin, a.j.src + a.pos, n, hipMemcpyDeviceToHost, stream);
            hipError_t e3 = hipEventRecord(s.t1, stream);
            if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess)
                fail(std::string("ns_io: device-to-host copy: ") + hipGetErrorString(e2 != hipSuccess ? e2 : e1 != hipSuccess ? e1 : e3));
            {
                std::lock_guard<std::mutex> g(mu);
                copied_jobs.push_back(WriteJob{si, a.j.sink, a.j.file_off + a.pos, n, a.j.slot});
            }
            cv_write.notify_one();
            a.pos += n;
            if (a.pos >= a.j.n) act.erase(act.begin() + (ptrdiff_t)rr); else ++rr;
        }
    }
    // A writer thread alternates between two duties: (1) wait for the copy event of the oldest issued slice — the slice has then left
    // the device (its result slot may be reused) and is READY —, (2) write a ready slice whose file no other writer is in.
    void write_loop() {
        hipError_t e = hipSetDevice(device); (void)e;
        for (;;) {
            WriteJob j;
            bool ready = false;
            {
                std::unique_lock<std::mutex> g(mu);
                std::deque<WriteJob>::iterator it;
                cv_write.wait(g, [&] {
                    for (it = ready_jobs.begin(); it != ready_jobs.end(); ++it) if (!it->sink->writing) return true;
                    return !copied_jobs.empty() || (stop && ready_jobs.empty());
                });
                if (it != ready_jobs.end()) { j = *it; ready_jobs.erase(it); j.sink->writing = true; ready = true; }
                else if (!copied_jobs.empty()) { j = copied_jobs.front(); copied_jobs.pop_front(); }
                else return;
            }
            Slice &s = slices[j.slice];
            if (!ready) {
                float ms = 0;
                hipError_t e1 = hipEventSynchronize(s.t1);
                if (e1 == hipSuccess) e1 = hipEventElapsedTime(&ms, s.t0, s.t1);
                if (e1 != hipSuccess) fail(std::string("ns_io: copy event: ") + hipGetErrorString(e1));
                {
                    std::lock_guard<std::mutex> g(mu);
                    --slot_pending[j.slot]; dma_ms += ms; bytes += j.n;
                    ready_jobs.push_back(j);
                }
                cv_idle.notify_all(); cv_write.notify_one();
                continue;
            }
            const auto t0 = std::chrono::steady_clock::now();
            if (j.sink->fd >= 0 && !j.sink->err.load()) {
                size_t done = 0;
                while (done < j.n) {
                    const ssize_t w = pwrite(j.sink->fd, s.pin + done, j.n - done, (off_t)(j.file_off + done));
                    if (w < 0) { if (errno == EINTR) continue; int z = 0; j.sink->err.compare_exchange_strong(z, errno ? errno : EIO); break; }
                    if (w == 0) { int z = 0; j.sink->err.compare_exchange_strong(z, ENOSPC); break; }
                    done += (size_t)w;
                }
            }
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            {
                // `written` is published LAST and under the lock: ns_sink_close -> wait_sink tests written == queued under the same
                // lock and then deletes the sink, so nothing may touch it after that update becomes visible
                std::lock_guard<std::mutex> g(mu);
                free_slices.push_back(j.slice); --jobs_open; write_s += dt;
                j.sink->writing = false;
                j.sink->written += j.n;
            }
            cv_free.notify_one(); cv_idle.notify_all(); cv_write.notify_one();
        }
    }
};
