"""BoxCoder: boxes <-> regression deltas (reference modeling/box_coder.py:7-95).

encode: (dx, dy, dw, dh) = (wx*(gx-ex)/ew, wy*(gy-ey)/eh, ww*log(gw/ew), wh*log(gh/eh)) with the
+1 pixel widths; decode inverts it, clamping dw/dh at log(1000/16) and producing inclusive corners
(x2 = cx + 0.5*w - 1).  Pinned by tests/golden/box_coder_reference_tests.npz (the reference's
tests/test_box_coder.py vectors)."""
import math

import torch

TO_REMOVE = 1


class BoxCoder(object):
    def __init__(self, weights, bbox_xform_clip=math.log(1000. / 16)):
        self.weights = weights
        self.bbox_xform_clip = bbox_xform_clip

    @staticmethod
    def _center_form(boxes):
        w = boxes[..., 2] - boxes[..., 0] + TO_REMOVE
        h = boxes[..., 3] - boxes[..., 1] + TO_REMOVE
        return boxes[..., 0] + 0.5 * w, boxes[..., 1] + 0.5 * h, w, h

    def encode(self, reference_boxes, proposals):
        """deltas that move `proposals` onto `reference_boxes` (both [..., 4] xyxy)."""
        ex, ey, ew, eh = self._center_form(proposals)
        gx, gy, gw, gh = self._center_form(reference_boxes)
        wx, wy, ww, wh = self.weights
        return torch.stack((wx * (gx - ex) / ew, wy * (gy - ey) / eh,
                            ww * torch.log(gw / ew), wh * torch.log(gh / eh)), dim=-1)

    def decode(self, rel_codes, boxes):
        """apply deltas `rel_codes` [n, 4*k] to `boxes` [n, 4] -> [n, 4*k]."""
        boxes = boxes.to(rel_codes.dtype)
        cx, cy, w, h = (t[:, None] for t in self._center_form(boxes))
        wx, wy, ww, wh = self.weights
        dx = rel_codes[:, 0::4] / wx
        dy = rel_codes[:, 1::4] / wy
        dw = torch.clamp(rel_codes[:, 2::4] / ww, max=self.bbox_xform_clip)
        dh = torch.clamp(rel_codes[:, 3::4] / wh, max=self.bbox_xform_clip)
        pcx, pcy = dx * w + cx, dy * h + cy
        pw, ph = torch.exp(dw) * w, torch.exp(dh) * h
        out = torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw - 1, pcy + 0.5 * ph - 1), dim=2)
        return out.reshape(rel_codes.shape[0], -1)
