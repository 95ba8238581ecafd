#!/usr/bin/env python3
"""gmmbin/gmm-latgen-faster.cc's command line over the library; see tools/latgen_faster.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import latgen_faster  # noqa: E402


def main(argv=None):
    return latgen_faster.main(argv, kind="gmm")


if __name__ == "__main__":
    sys.exit(main())
