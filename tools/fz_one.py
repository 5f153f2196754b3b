import sys, os
sys.path.insert(0, "tests/perf"); sys.path.insert(0, "tests")
import fuzz_parity
print(os.environ.get("TAG"), fuzz_parity.one_case(7060))
