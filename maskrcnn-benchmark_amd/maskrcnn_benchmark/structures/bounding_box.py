"""BoxList — the box container of the model surface (reference structures/bounding_box.py:9-255).

Same public behaviour: boxes `[n,4]` fp32 with an image size `(width, height)`, a mode in
{"xyxy", "xywh"}, the inclusive-pixel convention (TO_REMOVE = 1) and a dict of per-box extra
fields that follows the boxes through indexing / resizing / device moves.

One addition used by the sync-free training path of this framework: a BoxList may carry a boolean
field "valid".  Padded (fixed-length) proposal sets use it instead of being compacted with
`nonzero`, which would force a device->host round trip per image per iteration.
"""
import torch

FLIP_LEFT_RIGHT = 0
FLIP_TOP_BOTTOM = 1
TO_REMOVE = 1


class BoxList(object):
    def __init__(self, bbox, image_size, mode="xyxy"):
        device = bbox.device if isinstance(bbox, torch.Tensor) else torch.device("cpu")
        bbox = torch.as_tensor(bbox, dtype=torch.float32, device=device)
        if bbox.ndimension() != 2:
            raise ValueError("bbox should have 2 dimensions, got {}".format(bbox.ndimension()))
        if bbox.size(-1) != 4:
            raise ValueError("last dimension of bbox should have a size of 4, got {}".format(bbox.size(-1)))
        if mode not in ("xyxy", "xywh"):
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        self.bbox = bbox
        self.size = image_size  # (image_width, image_height)
        self.mode = mode
        self.extra_fields = {}

    # ---- extra fields
    def add_field(self, field, field_data):
        self.extra_fields[field] = field_data

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return list(self.extra_fields.keys())

    def _copy_extra_fields(self, other):
        for k, v in other.extra_fields.items():
            self.extra_fields[k] = v

    # ---- representation changes
    def _split_into_xyxy(self):
        a, b, c, d = self.bbox.split(1, dim=-1)
        if self.mode == "xyxy":
            return a, b, c, d
        # xywh: width/height are inclusive pixel counts
        return a, b, a + (c - TO_REMOVE).clamp(min=0), b + (d - TO_REMOVE).clamp(min=0)

    def convert(self, mode):
        if mode not in ("xyxy", "xywh"):
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        if mode == self.mode:
            return self
        x1, y1, x2, y2 = self._split_into_xyxy()
        if mode == "xyxy":
            data = torch.cat((x1, y1, x2, y2), dim=-1)
        else:
            data = torch.cat((x1, y1, x2 - x1 + TO_REMOVE, y2 - y1 + TO_REMOVE), dim=-1)
        out = BoxList(data, self.size, mode=mode)
        out._copy_extra_fields(self)
        return out

    def resize(self, size, *args, **kwargs):
        """Rescale to an image of `size` (width, height); extra fields that know how to resize
        (masks) are resized too."""
        rw, rh = (float(s) / float(o) for s, o in zip(size, self.size))
        if rw == rh:
            out = BoxList(self.bbox * rw, size, mode=self.mode)
        else:
            x1, y1, x2, y2 = self._split_into_xyxy()
            out = BoxList(torch.cat((x1 * rw, y1 * rh, x2 * rw, y2 * rh), dim=-1), size, mode="xyxy")
        for k, v in self.extra_fields.items():
            if not isinstance(v, torch.Tensor):
                v = v.resize(size, *args, **kwargs)
            out.add_field(k, v)
        return out if rw == rh else out.convert(self.mode)

    def transpose(self, method):
        if method not in (FLIP_LEFT_RIGHT, FLIP_TOP_BOTTOM):
            raise NotImplementedError("Only FLIP_LEFT_RIGHT and FLIP_TOP_BOTTOM implemented")
        W, H = self.size
        x1, y1, x2, y2 = self._split_into_xyxy()
        if method == FLIP_LEFT_RIGHT:
            x1, x2 = W - x2 - TO_REMOVE, W - x1 - TO_REMOVE
        else:
            y1, y2 = H - y2, H - y1  # (sic) the reference's vertical flip has no TO_REMOVE
        out = BoxList(torch.cat((x1, y1, x2, y2), dim=-1), self.size, mode="xyxy")
        for k, v in self.extra_fields.items():
            if not isinstance(v, torch.Tensor):
                v = v.transpose(method)
            out.add_field(k, v)
        return out.convert(self.mode)

    def crop(self, box):
        """Crop to the rectangle `box` = (x1, y1, x2, y2); coordinates become relative to it."""
        x1, y1, x2, y2 = self._split_into_xyxy()
        w, h = box[2] - box[0], box[3] - box[1]
        data = torch.cat(((x1 - box[0]).clamp(min=0, max=w), (y1 - box[1]).clamp(min=0, max=h),
                          (x2 - box[0]).clamp(min=0, max=w), (y2 - box[1]).clamp(min=0, max=h)), dim=-1)
        out = BoxList(data, (w, h), mode="xyxy")
        for k, v in self.extra_fields.items():
            if not isinstance(v, torch.Tensor):
                v = v.crop(box)
            out.add_field(k, v)
        return out.convert(self.mode)

    # ---- tensor-like
    def to(self, device):
        out = BoxList(self.bbox.to(device), self.size, self.mode)
        for k, v in self.extra_fields.items():
            if hasattr(v, "to"):
                v = v.to(device)
            out.add_field(k, v)
        return out

    def __getitem__(self, item):
        out = BoxList(self.bbox[item], self.size, self.mode)
        for k, v in self.extra_fields.items():
            out.add_field(k, v[item])
        return out

    def __len__(self):
        return self.bbox.shape[0]

    def clip_to_image(self, remove_empty=True):
        W, H = self.size
        self.bbox[:, 0].clamp_(min=0, max=W - TO_REMOVE)
        self.bbox[:, 1].clamp_(min=0, max=H - TO_REMOVE)
        self.bbox[:, 2].clamp_(min=0, max=W - TO_REMOVE)
        self.bbox[:, 3].clamp_(min=0, max=H - TO_REMOVE)
        if remove_empty:
            b = self.bbox
            return self[(b[:, 3] > b[:, 1]) & (b[:, 2] > b[:, 0])]
        return self

    def area(self):
        b = self.bbox
        if self.mode == "xyxy":
            return (b[:, 2] - b[:, 0] + TO_REMOVE) * (b[:, 3] - b[:, 1] + TO_REMOVE)
        return b[:, 2] * b[:, 3]

    def copy_with_fields(self, fields, skip_missing=False):
        out = BoxList(self.bbox, self.size, self.mode)
        if not isinstance(fields, (list, tuple)):
            fields = [fields]
        for f in fields:
            if self.has_field(f):
                out.add_field(f, self.get_field(f))
            elif not skip_missing:
                raise KeyError("Field '{}' not found in {}".format(f, self))
        return out

    def __repr__(self):
        return "{}(num_boxes={}, image_width={}, image_height={}, mode={})".format(
            self.__class__.__name__, len(self), self.size[0], self.size[1], self.mode)
