"""reference utils/imsitu_loader.py: (img_name, img, verb, labels) items from an annotation dict and an image folder."""
import os

import torch.utils.data as data
from PIL import Image


class imsitu_loader(data.Dataset):
    def __init__(self, img_dir, train_json, encoder, transform=None):
        self.img_dir, self.train_json, self.encoder, self.transform = img_dir, train_json, encoder, transform
        self.imgs_names = list(train_json.keys())

    def __getitem__(self, index):
        name = self.imgs_names[index]
        img = Image.open(os.path.join(self.img_dir, name)).convert('RGB')
        if self.transform is not None:
            img = self.transform(img)
        verb, labels = self.encoder.encode(self.train_json[name])
        return name, img, verb, labels

    def __len__(self):
        return len(self.train_json)
