ScalarFloat = float
