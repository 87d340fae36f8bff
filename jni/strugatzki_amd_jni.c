/*
 * strugatzki_amd_jni.c -- JNI shim between de.sciss.strugatzki.impl.Native (scala/) and the C ABI of
 * libstrugatzki_amd.so (include/strugatzki_amd.h).  One wrapper per processor of the reference:
 *
 *   Native.runCorrelation      -> sgz_run_correlation_xml / _multi   (impl/FeatureCorrelationImpl.scala:32-412)
 *   Native.runSegmentation     -> sgz_run_segmentation_xml           (impl/FeatureSegmentationImpl.scala:31-142)
 *   Native.runSelfSimilarity   -> sgz_run_selfsimilarity_xml         (impl/SelfSimilarityImpl.scala:31-180)
 *   Native.selfSimilarityRows  -> sgz_run_selfsimilarity_rows_xml, bands of similarities (the JVM colours them: IntensityPalette)
 *   Native.runExtraction       -> sgz_run_extraction_xml             (impl/FeatureExtractionImpl.scala:31-78)
 *   Native.runCrossSimilarity  -> sgz_run_crosssimilarity_xml        (impl/CrossSimilarityImpl.scala:32-173)
 *   Native.runFeatureStats     -> sgz_run_feature_stats              (impl/FeatureStatsImpl.scala:30-54)
 *
 * Build where a JDK is installed (none exists in the image the library is built in, hence the guard):
 *   cc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude jni/strugatzki_amd_jni.c \
 *      -Lstrugatzki_amd/lib -lstrugatzki_amd -o libstrugatzki_amd_jni.so
 * Without <jni.h> the file compiles to an empty object, so that it can sit in any build.
 */
#if defined(__has_include)
#if __has_include(<jni.h>)
#define SGZ_HAVE_JNI 1
#endif
#endif

#ifdef SGZ_HAVE_JNI
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "strugatzki_amd.h"

#define NATIVE(name) Java_de_sciss_strugatzki_impl_Native_00024_##name

/* Native.Progress.apply(p: Double): Boolean -- false aborts the run (ProcessorImpl.checkAborted) */
typedef struct {
    JNIEnv *env;
    jobject obs;
    jmethodID apply;
} progress_t;

static int on_progress(double p, void *user)
{
    progress_t *c = (progress_t *)user;
    if (!c->obs) return 1;
    jboolean go = (*c->env)->CallBooleanMethod(c->env, c->obs, c->apply, (jdouble)p);
    if ((*c->env)->ExceptionCheck(c->env)) return 0; /* the pending exception surfaces when the call returns */
    return go ? 1 : 0;
}

static progress_t make_progress(JNIEnv *env, jobject obs)
{
    progress_t c = {env, obs, NULL};
    if (obs) c.apply = (*env)->GetMethodID(env, (*env)->GetObjectClass(env, obs), "apply", "(D)Z");
    return c;
}

/* SGZ_E_ARG -> IllegalArgumentException (the reference's require(...)), SGZ_E_IO -> IOException,
 * "aborted" -> the Scala side turns the IllegalStateException with that message into Processor.Aborted() */
static void throw_for(JNIEnv *env, int rc, const char *msg)
{
    if ((*env)->ExceptionCheck(env)) return;
    const char *cls = rc == SGZ_E_ARG  ? "java/lang/IllegalArgumentException"
                      : rc == SGZ_E_IO ? "java/io/IOException"
                                       : "java/lang/IllegalStateException";
    (*env)->ThrowNew(env, (*env)->FindClass(env, cls), msg && *msg ? msg : "strugatzki_amd failure");
}

static sgz_context *ctx_of(jlong h) { return (sgz_context *)(intptr_t)h; }

JNIEXPORT jlong JNICALL NATIVE(create)(JNIEnv *env, jobject self, jint device)
{
    (void)self;
    sgz_context *ctx = NULL;
    const int rc = sgz_create((int)device, NULL, &ctx);
    if (rc != SGZ_OK) throw_for(env, rc, "no usable GPU");
    return (jlong)(intptr_t)ctx;
}

JNIEXPORT void JNICALL NATIVE(destroy)(JNIEnv *env, jobject self, jlong h)
{
    (void)env;
    (void)self;
    sgz_destroy(ctx_of(h));
}

/* contexts: one handle per GPU; one handle = the single-GPU call */
JNIEXPORT jstring JNICALL NATIVE(runCorrelation)(JNIEnv *env, jobject self, jlongArray contexts, jstring cfg, jobject obs)
{
    (void)self;
    const jsize n = (*env)->GetArrayLength(env, contexts);
    if (n < 1) {
        throw_for(env, SGZ_E_ARG, "no context");
        return NULL;
    }
    jlong *hs = (*env)->GetLongArrayElements(env, contexts, NULL);
    sgz_context **ctxs = (sgz_context **)malloc(sizeof(sgz_context *) * (size_t)n);
    for (jsize i = 0; i < n; i++) ctxs[i] = ctx_of(hs[i]);
    (*env)->ReleaseLongArrayElements(env, contexts, hs, JNI_ABORT);
    const char *xml = (*env)->GetStringUTFChars(env, cfg, NULL);
    progress_t cb = make_progress(env, obs);
    char *out = NULL;
    const int rc = n == 1 ? sgz_run_correlation_xml(ctxs[0], xml, on_progress, &cb, &out)
                          : sgz_run_correlation_xml_multi(ctxs, (int)n, xml, on_progress, &cb, &out);
    (*env)->ReleaseStringUTFChars(env, cfg, xml);
    free(ctxs);
    if (rc != SGZ_OK) {
        throw_for(env, rc, sgz_host_last_error());
        return NULL;
    }
    jstring res = (*env)->NewStringUTF(env, out);
    sgz_free_string(out);
    return res;
}

JNIEXPORT jstring JNICALL NATIVE(runSegmentation)(JNIEnv *env, jobject self, jlong h, jstring cfg, jobject obs)
{
    (void)self;
    const char *xml = (*env)->GetStringUTFChars(env, cfg, NULL);
    progress_t cb = make_progress(env, obs);
    char *out = NULL;
    const int rc = sgz_run_segmentation_xml(ctx_of(h), xml, on_progress, &cb, &out);
    (*env)->ReleaseStringUTFChars(env, cfg, xml);
    if (rc != SGZ_OK) {
        throw_for(env, rc, sgz_host_last_error());
        return NULL;
    }
    jstring res = (*env)->NewStringUTF(env, out);
    sgz_free_string(out);
    return res;
}

typedef int (*xml_run_fn)(sgz_context *, const char *, sgz_progress_fn, void *);

static void run_unit(JNIEnv *env, jlong h, jstring cfg, jobject obs, xml_run_fn fn)
{
    const char *xml = (*env)->GetStringUTFChars(env, cfg, NULL);
    progress_t cb = make_progress(env, obs);
    const int rc = fn(ctx_of(h), xml, on_progress, &cb);
    (*env)->ReleaseStringUTFChars(env, cfg, xml);
    if (rc != SGZ_OK) throw_for(env, rc, sgz_host_last_error());
}

JNIEXPORT void JNICALL NATIVE(runSelfSimilarity)(JNIEnv *env, jobject self, jlong h, jstring cfg, jobject obs)
{
    (void)self;
    run_unit(env, h, cfg, obs, sgz_run_selfsimilarity_xml);
}

JNIEXPORT void JNICALL NATIVE(runExtraction)(JNIEnv *env, jobject self, jlong h, jstring cfg, jobject obs)
{
    (void)self;
    run_unit(env, h, cfg, obs, sgz_run_extraction_xml);
}

JNIEXPORT void JNICALL NATIVE(runCrossSimilarity)(JNIEnv *env, jobject self, jlong h, jstring cfg, jobject obs)
{
    (void)self;
    run_unit(env, h, cfg, obs, sgz_run_crosssimilarity_xml);
}

/* FeatureStats: paths in, [min0, max0, min1, max1, ...] out */
JNIEXPORT jdoubleArray JNICALL NATIVE(runFeatureStats)(JNIEnv *env, jobject self, jlong h, jobjectArray paths, jobject obs)
{
    (void)self;
    const jsize n = (*env)->GetArrayLength(env, paths);
    const char **cs = (const char **)calloc((size_t)(n > 0 ? n : 1), sizeof(char *));
    jstring *js = (jstring *)calloc((size_t)(n > 0 ? n : 1), sizeof(jstring));
    for (jsize i = 0; i < n; i++) {
        js[i] = (jstring)(*env)->GetObjectArrayElement(env, paths, i);
        cs[i] = (*env)->GetStringUTFChars(env, js[i], NULL);
    }
    progress_t cb = make_progress(env, obs);
    double minmax[2 * SGZ_MAX_CHANNELS];
    int numCh = 0;
    const int rc = sgz_run_feature_stats(ctx_of(h), cs, (int)n, on_progress, &cb, minmax, SGZ_MAX_CHANNELS, &numCh);
    for (jsize i = 0; i < n; i++) (*env)->ReleaseStringUTFChars(env, js[i], cs[i]);
    free(cs);
    free(js);
    if (rc != SGZ_OK) {
        throw_for(env, rc, sgz_host_last_error());
        return NULL;
    }
    jdoubleArray res = (*env)->NewDoubleArray(env, 2 * numCh);
    (*env)->SetDoubleArrayRegion(env, res, 0, 2 * numCh, minmax);
    return res;
}

/* Native.selfSimilarityRows: rows [xBegin, xEnd) of the matrix of a <selfsimilarity> document into a direct
 * FloatBuffer of at least (xEnd - xBegin) * extent floats (may be null when xBegin == xEnd); returns the extent. */
JNIEXPORT jint JNICALL NATIVE(selfSimilarityRows)(JNIEnv *env, jobject self, jlong h, jstring cfg, jint xBegin, jint xEnd,
                                                 jobject sims)
{
    (void)self;
    int ext = 0;
    const char *xml = (*env)->GetStringUTFChars(env, cfg, NULL);
    if (!xml) return 0;
    float *out = sims ? (float *)(*env)->GetDirectBufferAddress(env, sims) : NULL;
    int rc = SGZ_OK;
    if (xEnd > xBegin) {  /* the buffer must hold the band: ask for the extent first */
        rc = sgz_run_selfsimilarity_rows_xml(ctx_of(h), xml, 0, 0, NULL, &ext);
        if (rc == SGZ_OK && (!out || (*env)->GetDirectBufferCapacity(env, sims) < (jlong)(xEnd - xBegin) * ext)) {
            (*env)->ReleaseStringUTFChars(env, cfg, xml);
            throw_for(env, SGZ_E_ARG, "a direct buffer of (xEnd - xBegin) * extent floats is required");
            return 0;
        }
    }
    if (rc == SGZ_OK) rc = sgz_run_selfsimilarity_rows_xml(ctx_of(h), xml, (int)xBegin, (int)xEnd, out, &ext);
    (*env)->ReleaseStringUTFChars(env, cfg, xml);
    if (rc != SGZ_OK) {
        const char *msg = sgz_host_last_error();
        throw_for(env, rc, msg && *msg ? msg : sgz_last_error(ctx_of(h)));
        return 0;
    }
    return (jint)ext;
}

#endif /* SGZ_HAVE_JNI */
