class ToTensor:
    def __call__(self, img):
        raise NotImplementedError("torchvision stand-in: image decoding is outside what the fixtures pin")


class ToPILImage(ToTensor):
    pass
