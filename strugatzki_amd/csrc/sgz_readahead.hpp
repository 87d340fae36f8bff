// Read-ahead of audio files for the loops that load a database (host code, no GPU calls).
#pragma once

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <exception>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "strugatzki_amd.hpp"

namespace strugatzki {

// Files read ahead of their use: a few threads read the next files of `paths` (readAudioFilePcm) while the caller
// takes them in order -- consume(i, file) runs on the calling thread only, which is where the device is used.  At
// most kWindow files are held at a time.  A file that cannot be read throws from here when its turn comes, as a
// plain loop would; whatever consume throws passes through after the readers have stopped.
template <class Consume>
inline void forEachFileReadAhead(const std::vector<std::string> &paths, Consume consume)
{
    constexpr size_t kWindow = 16;
    const size_t n = paths.size();
    if (n == 0) return;
    struct Slot {
        std::unique_ptr<PcmData> data;
        std::exception_ptr error;
        bool ready = false;
    };
    std::vector<Slot> slots(n);
    std::mutex mu;
    std::condition_variable cv;
    size_t consumed = 0;
    bool stop = false;
    std::atomic<size_t> next{0};
    auto reader = [&] {
        for (;;) {
            const size_t i = next++;
            if (i >= n) return;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || i < consumed + kWindow; });
                if (stop) return;
            }
            std::unique_ptr<PcmData> data;
            std::exception_ptr error;
            try {
                data.reset(new PcmData(readAudioFilePcm(paths[i])));
            } catch (...) {
                error = std::current_exception();
            }
            std::lock_guard<std::mutex> lk(mu);
            slots[i].data = std::move(data);
            slots[i].error = error;
            slots[i].ready = true;
            cv.notify_all();
        }
    };
    std::vector<std::thread> readers;
    struct Join {
        std::vector<std::thread> &t;
        std::mutex &mu;
        std::condition_variable &cv;
        bool &stop;
        ~Join()
        {
            {
                std::lock_guard<std::mutex> lk(mu);
                stop = true;
            }
            cv.notify_all();
            for (auto &th : t) th.join();
        }
    } join{readers, mu, cv, stop};
    const size_t numReaders = std::min<size_t>(n, std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency())));
    for (size_t t = 0; t < numReaders; t++) readers.emplace_back(reader);
    for (size_t i = 0; i < n; i++) {
        std::unique_ptr<PcmData> data;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return slots[i].ready; });
            if (slots[i].error) std::rethrow_exception(slots[i].error);
            data = std::move(slots[i].data);
        }
        consume(i, *data);
        {
            std::lock_guard<std::mutex> lk(mu);
            consumed = i + 1;
        }
        cv.notify_all();
    }
}

}  // namespace strugatzki
