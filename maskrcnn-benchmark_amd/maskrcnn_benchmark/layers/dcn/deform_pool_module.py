"""DeformRoIPooling modules (reference layers/dcn/deform_pool_module.py:6-150)."""
from torch import nn

from .deform_pool_func import deform_roi_pooling


def _fc_stack(in_features, hidden, out_features, depth):
    layers, c = [], in_features
    for _ in range(depth):
        layers += [nn.Linear(c, hidden), nn.ReLU(inplace=True)]
        c = hidden
    layers.append(nn.Linear(c, out_features))
    return layers


class DeformRoIPooling(nn.Module):
    def __init__(self, spatial_scale, out_size, out_channels, no_trans, group_size=1, part_size=None,
                 sample_per_part=4, trans_std=.0):
        super(DeformRoIPooling, self).__init__()
        self.spatial_scale = spatial_scale
        self.out_size = out_size
        self.out_channels = out_channels
        self.no_trans = no_trans
        self.group_size = group_size
        self.part_size = out_size if part_size is None else part_size
        self.sample_per_part = sample_per_part
        self.trans_std = trans_std

    def _pool(self, data, rois, offset, no_trans):
        return deform_roi_pooling(data, rois, offset, self.spatial_scale, self.out_size,
                                  self.out_channels, no_trans, self.group_size, self.part_size,
                                  self.sample_per_part, self.trans_std)

    def forward(self, data, rois, offset):
        if self.no_trans:
            offset = data.new_empty(0)
        return self._pool(data, rois, offset, self.no_trans)


class DeformRoIPoolingPack(DeformRoIPooling):
    def __init__(self, spatial_scale, out_size, out_channels, no_trans, group_size=1, part_size=None,
                 sample_per_part=4, trans_std=.0, deform_fc_channels=1024):
        super(DeformRoIPoolingPack, self).__init__(spatial_scale, out_size, out_channels, no_trans,
                                                   group_size, part_size, sample_per_part, trans_std)
        self.deform_fc_channels = deform_fc_channels
        if not no_trans:
            feat = self.out_size * self.out_size
            self.offset_fc = nn.Sequential(*_fc_stack(feat * self.out_channels, deform_fc_channels,
                                                      feat * 2, depth=2))
            self.offset_fc[-1].weight.data.zero_()
            self.offset_fc[-1].bias.data.zero_()

    def _offsets(self, data, rois):
        n = rois.shape[0]
        x = self._pool(data, rois, data.new_empty(0), True)
        return x, self.offset_fc(x.view(n, -1)).view(n, 2, self.out_size, self.out_size)

    def forward(self, data, rois):
        assert data.size(1) == self.out_channels
        if self.no_trans:
            return self._pool(data, rois, data.new_empty(0), True)
        _, offset = self._offsets(data, rois)
        return self._pool(data, rois, offset, self.no_trans)


class ModulatedDeformRoIPoolingPack(DeformRoIPoolingPack):
    def __init__(self, spatial_scale, out_size, out_channels, no_trans, group_size=1, part_size=None,
                 sample_per_part=4, trans_std=.0, deform_fc_channels=1024):
        super(ModulatedDeformRoIPoolingPack, self).__init__(
            spatial_scale, out_size, out_channels, no_trans, group_size, part_size, sample_per_part,
            trans_std, deform_fc_channels)
        if not no_trans:
            feat = self.out_size * self.out_size
            self.mask_fc = nn.Sequential(*(_fc_stack(feat * self.out_channels, deform_fc_channels,
                                                     feat, depth=1) + [nn.Sigmoid()]))
            self.mask_fc[2].weight.data.zero_()
            self.mask_fc[2].bias.data.zero_()

    def forward(self, data, rois):
        assert data.size(1) == self.out_channels
        if self.no_trans:
            return self._pool(data, rois, data.new_empty(0), True)
        n = rois.shape[0]
        x, offset = self._offsets(data, rois)
        mask = self.mask_fc(x.view(n, -1)).view(n, 1, self.out_size, self.out_size)
        return self._pool(data, rois, offset, self.no_trans) * mask
