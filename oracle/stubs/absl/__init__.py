"""Build-owned test stub: `absl` is imported by the reference's synthetic-data script for its command line only
(lib/datasets/synthetic.py:5-6, 273-282); nothing on the tested path calls it."""
