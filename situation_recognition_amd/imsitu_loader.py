"""Dataset with the item layout of reference utils/imsitu_loader.py -- (img_name, img, verb, labels) -- over an annotation
dict and an image folder.  The annotations are encoded once at construction (the reference re-runs the list-scanning
`encoder.encode` for every access), so `__getitem__` only decodes and transforms the image."""
import os

import torch.utils.data as data
from PIL import Image


class imsitu_loader(data.Dataset):
    def __init__(self, img_dir, train_json, encoder, transform=None):
        self.img_dir, self.train_json, self.encoder, self.transform = img_dir, train_json, encoder, transform
        self.imgs_names = list(train_json)
        self._encoded = [encoder.encode(train_json[n]) for n in self.imgs_names]     # [(verb id, labels [3,R])]

    def __len__(self):
        return len(self.imgs_names)

    def _image(self, name):
        with Image.open(os.path.join(self.img_dir, name)) as im:
            rgb = im.convert('RGB')
        return rgb if self.transform is None else self.transform(rgb)

    def __getitem__(self, index):
        name = self.imgs_names[index]
        verb, labels = self._encoded[index]
        return name, self._image(name), verb, labels
