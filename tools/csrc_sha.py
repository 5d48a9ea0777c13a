"""Print the sha16 of the kernel sources + ABI header (bench.py csrc_sha16): tools/final_measure.sh records it next to the PMC passes so
that bench.py only reports `roofline.traffic` from a table measured on the code it is running."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(bench.csrc_sha16())
