"""The reference's two matplotlib figure loggers (lib/loggers/loggers.py) are plotting helpers
that no script calls; out of scope.  The module exists so `train_image.py`'s
imports resolve."""
