"""TSNDataSet with the reference's constructor and item format (dataset.py:31-144):
a list file of `<dir> <num_frames> <label>` lines and one torch-saved 1-D feature
tensor per frame (`img_{:05d}.t7`).  The deterministic test-mode segment selection
(dataset.py:103-116 - the only sampler main.py uses, main.py:171-197) comes from the
C library and is bit-exact; the file loading itself is outside the timed path
(SURVEY.md 8f rank 2)."""
import os

import torch
import torch.utils.data as data

from . import _lib


class VideoRecord(object):
    def __init__(self, row):
        self._data = row

    @property
    def path(self):
        return self._data[0]

    @property
    def num_frames(self):
        return int(self._data[1])

    @property
    def label(self):
        return int(self._data[2])


class TSNDataSet(data.Dataset):
    def __init__(self, root_path, list_file, num_dataload, num_segments=3, new_length=1, modality='RGB',
                 image_tmpl='img_{:05d}.t7', transform=None, force_grayscale=False, random_shift=True,
                 test_mode=False):
        self.root_path, self.list_file = root_path, list_file
        self.num_segments, self.new_length, self.modality = num_segments, new_length, modality
        self.image_tmpl, self.transform = image_tmpl, transform
        self.random_shift, self.test_mode, self.num_dataload = random_shift, test_mode, num_dataload
        if modality in ('RGBDiff', 'RGBDiff2', 'RGBDiffplus'):
            self.new_length += 1                                         # dataset.py:47-48
        if modality == 'Flow':
            raise NotImplementedError("Flow features (x/y file pairs, dataset.py:62-66) are outside the TA3N hot path")
        if not test_mode:
            raise NotImplementedError("only test_mode=True sampling is used by main.py (main.py:171-197)")
        self._parse_list()

    def _parse_list(self):
        """dataset.py:69-74: the list is repeated and truncated to num_dataload entries."""
        rows = [VideoRecord(x.strip().split(' ')) for x in open(self.list_file)]
        n_repeat = self.num_dataload // len(rows)
        n_left = self.num_dataload % len(rows)
        self.video_list = rows * n_repeat + rows[:n_left]

    def _get_test_indices(self, record):
        return _lib.segment_indices(record.num_frames, self.num_segments, self.new_length)

    def _load_feature(self, directory, idx):
        return [torch.load(os.path.join(directory, self.image_tmpl.format(idx)))]

    def __getitem__(self, index):
        record = self.video_list[index]
        frames = []
        for seg_ind in self._get_test_indices(record):                   # dataset.py:128-144
            p = int(seg_ind)
            for _ in range(self.new_length):
                frames.extend(self._load_feature(record.path, p))
                if p < record.num_frames:
                    p += 1
        return torch.stack(frames), record.label

    def __len__(self):
        return len(self.video_list)
