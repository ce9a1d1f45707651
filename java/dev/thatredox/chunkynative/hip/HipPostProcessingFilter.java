package dev.thatredox.chunkynative.hip;

import se.llbit.chunky.renderer.postprocessing.PostProcessingFilter;
import se.llbit.chunky.resources.BitmapImage;
import se.llbit.util.TaskTracker;

/**
 * GPU tone mapping on the HIP library: stands in for a Chunky post-processing filter under that
 * filter's own name, description and id, like ImposterCombinationGpuPostProcessingFilter does for
 * the OpenCL build (tonemap/ImposterCombinationGpuPostProcessingFilter.java:10-29,
 * tonemap/GpuPostProcessingFilter.java:14-82).  Registration (ChunkyCl.java:60-72):
 *
 * <pre>
 *   register("GAMMA", Filter.GAMMA); register("TONEMAP1", Filter.TONEMAP1);
 *   register("TONEMAP2", Filter.ACES); register("TONEMAP3", Filter.HABLE);
 *   ...
 *   PostProcessingFilters.getPostProcessingFilterFromId(id).ifPresent(f ->
 *       PostProcessingFilters.addPostProcessingFilter(new HipPostProcessingFilter(f, type, ctx)));
 * </pre>
 *
 * NOT compiled in this repository (no JDK / chunky-core here; see INTEGRATION.md).
 */
public class HipPostProcessingFilter implements PostProcessingFilter {
    public enum Filter {
        GAMMA(0), TONEMAP1(1), ACES(2), HABLE(3);

        public final int id;

        Filter(int id) {
            this.id = id;
        }
    }

    private final String name, description, id;
    private final Filter filter;
    private final long ctx;

    public HipPostProcessingFilter(PostProcessingFilter imposter, Filter filter, long ctx) {
        this.name = imposter.getName();
        this.description = imposter.getDescription();
        this.id = imposter.getId();
        this.filter = filter;
        this.ctx = ctx;
    }

    @Override
    public void processFrame(int width, int height, double[] input, BitmapImage output, double exposure,
                             TaskTracker.Task task) {
        // one blocking call: upload of the sample buffer, the `filter` kernel, read-back of the ARGB words
        HipNative.filterFrame(ctx, width, height, exposure, input, output.data, filter.id);
    }

    @Override
    public String getName() {
        return name;
    }

    @Override
    public String getDescription() {
        return description;
    }

    @Override
    public String getId() {
        return id;
    }
}
