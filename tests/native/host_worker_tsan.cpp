// HostWorker (csrc/internal.hpp) under ThreadSanitizer: one producer, jobs handed over and waited for while the "hot"
// count of live IPA states goes up and down -- the bounded spin window, the condition-variable fallback and shutdown.
#include <chrono>
#include <cstdio>
#include <thread>

#include "internal.hpp"

int main() {
    long total = 0;
    {
        halo::HostWorker w;
        for (int round = 0; round < 3; ++round) {
            w.add_hot(1);
            w.add_hot(1);  // two live states
            for (int i = 0; i < 2000; ++i) {
                long local = 0;
                w.submit([&local, i] { for (int k = 0; k < 50; ++k) local += i ^ k; });
                w.wait();
                total += local;
                if (i % 500 == 499) std::this_thread::sleep_for(std::chrono::milliseconds(2));  // longer than the spin window: the thread parks
            }
            w.add_hot(-1);  // one state destroyed: the other keeps the worker hot
            for (int i = 0; i < 200; ++i) { long local = 0; w.submit([&local] { local = 7; }); w.wait(); total += local; }
            w.add_hot(-1);
            for (int i = 0; i < 50; ++i) { long local = 0; w.submit([&local] { local = 1; }); w.wait(); total += local; }  // cold: every job through the condition variable
        }
    }  // destructor joins
    std::printf("ok %ld\n", total);
    return 0;
}
