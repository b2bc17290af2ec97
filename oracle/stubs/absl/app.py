def run(*a, **k):
    raise RuntimeError("absl.app stub: the command-line entry point is not part of the tested path")
