// The order checker's own logic (hisstools_library_amd/csrc/hcv_order_check.h), without a GPU: vector clocks over made-up stream and event
// handles.  Built and run by tests/test_order_check_unit.py; prints "ok <name>" per case and exits 0.
#include "hcv_order_check.h"

#include <cstdio>
#include <cstdlib>

using hcv::OrderCheck;

static hipStream_t S(int k) { return reinterpret_cast<hipStream_t>(static_cast<uintptr_t>(0x1000 + 16 * k)); }
static hipEvent_t E(int k) { return reinterpret_cast<hipEvent_t>(static_cast<uintptr_t>(0x9000 + 16 * k)); }
static int fails = 0;
#define CHECK(name, cond)                                         \
    do                                                            \
    {                                                             \
        if (cond) std::printf("ok %s\n", name);                   \
        else { std::printf("FAILED %s\n", name); fails++; }      \
    } while (0)

int main()
{
    setenv("HCV_ORDER_CHECK", "1", 1);
    static int bufA, bufB;
    {
        OrderCheck oc;
        oc.add_stream(S(0), "a");
        oc.add_stream(S(1), "b");
        // program order on one stream
        CHECK("same stream: write then read", oc.access(S(0), &bufA, 0, 8, 0, true, "w") == 0 && oc.access(S(0), &bufA, 0, 8, 0, false, "r") == 0);
        // another stream reads what the first wrote, with no event between them
        CHECK("cross stream read without an event is reported", oc.access(S(1), &bufA, 4, 6, 0, false, "r") == 1);
        // ... and with one
        oc.access(S(0), &bufB, 0, 8, 0, true, "w");
        oc.record(E(0), S(0));
        oc.wait(S(1), E(0));
        CHECK("record + wait orders the read", oc.access(S(1), &bufB, 0, 8, 0, false, "r") == 0);
        // the writer comes back without having waited for that reader
        CHECK("write over an unordered read is reported", oc.access(S(0), &bufB, 2, 3, 0, true, "w") == 1);
        // two readers never conflict
        CHECK("read beside read is fine", oc.access(S(0), &bufA, 0, 8, 0, false, "r") == 0);
        // disjoint elements never conflict
        oc.access(S(0), &bufA, 100, 110, 0, true, "w");
        CHECK("disjoint ranges", oc.access(S(1), &bufA, 110, 120, 0, true, "w") == 0);
    }
    {
        // an event recorded BEFORE the access does not cover it
        OrderCheck oc;
        oc.add_stream(S(0), "a");
        oc.add_stream(S(1), "b");
        oc.record(E(1), S(0));
        oc.access(S(0), &bufA, 0, 4, 0, true, "w");
        oc.wait(S(1), E(1));
        CHECK("an event recorded before the write does not order it", oc.access(S(1), &bufA, 0, 4, 0, false, "r") == 1);
    }
    {
        // rings: slot h and slot h + R are the same elements; a range may run past the end
        OrderCheck oc;
        oc.add_stream(S(0), "a");
        oc.add_stream(S(1), "b");
        oc.access(S(0), &bufA, 5, 6, 18, true, "w slot 5");
        CHECK("ring: the slot comes round", oc.access(S(1), &bufA, 23, 24, 18, false, "r slot 23 = 5") == 1);
        CHECK("ring: another slot", oc.access(S(1), &bufA, 24, 25, 18, false, "r slot 6") == 0);
        oc.access(S(0), &bufB, 16, 20, 18, true, "w slots 16, 17, 0, 1");
        CHECK("ring: a range that wraps", oc.access(S(1), &bufB, 0, 1, 18, false, "r slot 0") == 1 && oc.access(S(1), &bufB, 2, 3, 18, false, "r slot 2") == 0);
        CHECK("ring: negative positions", oc.access(S(1), &bufB, -1, 0, 18, false, "r slot 17") == 1);
    }
    {
        // transitivity through a third stream, a hand-over inside a launch, and the host's own waits
        OrderCheck oc;
        oc.add_stream(S(0), "a");
        oc.add_stream(S(1), "b");
        oc.add_stream(S(2), "c");
        oc.access(S(0), &bufA, 0, 4, 0, true, "w");
        oc.record(E(2), S(0));
        oc.wait(S(1), E(2));
        oc.record(E(3), S(1));
        oc.wait(S(2), E(3));
        CHECK("order is transitive", oc.access(S(2), &bufA, 0, 4, 0, true, "w") == 0);
        oc.access(S(0), &bufB, 0, 4, 0, true, "w");
        oc.meet(S(0), S(1));
        CHECK("a hand-over inside a launch (meet)", oc.access(S(1), &bufB, 0, 4, 0, false, "r") == 0);
        static int bufC;
        oc.access(S(2), &bufC, 0, 4, 0, true, "w");
        CHECK("without the host's wait", oc.access(S(0), &bufC, 0, 4, 0, false, "r") == 1);
        oc.access(S(2), &bufC, 0, 4, 0, true, "w again");
        oc.host_sync_stream(S(2));
        CHECK("after the host synchronized the stream", oc.access(S(0), &bufC, 0, 4, 0, true, "w") == 0);
        oc.host_sync_all();
        CHECK("after the host synchronized everything", oc.access(S(1), &bufC, 0, 4, 0, true, "w") == 0);
    }
    std::printf("violations counted: %lld\n", hcv::order_registry().violations.load());
    return fails ? 1 : 0;
}
