collect_ignore_glob = ["*"]  # scratch scripts (GPU micro-benchmarks, fuzzers), not tests
