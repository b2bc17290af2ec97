class MNIST:
    pass


class CIFAR10:
    pass
