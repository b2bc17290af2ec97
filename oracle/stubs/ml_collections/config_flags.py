"""Build-owned test stub (command-line helper of ml_collections; unused on the tested path)."""


def DEFINE_config_file(*a, **k):
    return None
