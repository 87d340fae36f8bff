/*
 *  Native.scala -- binding of libstrugatzki_amd.so (include/strugatzki_amd.h) through the JNI shim
 *  jni/strugatzki_amd_jni.c.  Drop into src/main/scala/de/sciss/strugatzki/impl/ of Strugatzki next to the
 *  *Impl.scala files of this directory, which replace the reference's classes of the same names.
 *
 *  Shipped as source: no JVM toolchain exists where the native library is built and tested; the tested surface
 *  is the C ABI these methods forward to.
 */
package de.sciss.strugatzki
package impl

import de.sciss.processor.Processor
import de.sciss.processor.impl.ProcessorImpl

private[strugatzki] object Native {
  System.loadLibrary("strugatzki_amd_jni") // links libstrugatzki_amd.so

  /** Receives the progress of a native run; `false` aborts it. */
  trait Progress { def apply(p: Double): Boolean }

  /** GPUs to use, e.g. `-Dstrugatzki.devices=0,1,2,3,4,5,6,7`; default: GPU 0. */
  def devices: Array[Int] =
    sys.props.get("strugatzki.devices").fold(Array(0))(_.split(',').map(_.trim.toInt))

  @native def create (device: Int): Long                                                      // sgz_create
  @native def destroy(ctx: Long): Unit                                                        // sgz_destroy
  /** One context: sgz_run_correlation_xml; several: sgz_run_correlation_xml_multi. Returns `<matches>`. */
  @native def runCorrelation    (contexts: Array[Long], configXML: String, obs: Progress): String
  @native def runSegmentation   (ctx: Long, configXML: String, obs: Progress): String       // `<breaks>`
  @native def runSelfSimilarity (ctx: Long, configXML: String, obs: Progress): Unit         // writes the PNG (gray)
  @native def runExtraction     (ctx: Long, configXML: String, obs: Progress): Unit
  @native def runCrossSimilarity(ctx: Long, configXML: String, obs: Progress): Unit
  /** Returns min0, max0, min1, max1, ... */
  @native def runFeatureStats   (ctx: Long, featureFiles: Array[String], obs: Progress): Array[Double]
  /** Rows [xBegin, xEnd) of the similarity matrix of a `<selfsimilarity>` document, [x - xBegin][y] (x = left), into
    * the direct buffer `sims` (no colours; inputs, norms and geometry are the library's affair); returns the matrix
    * extent.  `xBegin == xEnd`: the extent alone (`sims` may be null).  sgz_run_selfsimilarity_rows_xml. */
  @native def selfSimilarityRows(ctx: Long, configXML: String, xBegin: Int, xEnd: Int, sims: java.nio.FloatBuffer): Int

  /** Runs `fun` with one context per configured GPU and releases them afterwards. */
  def withContexts[A](devs: Array[Int])(fun: Array[Long] => A): A = {
    val ctx = devs.map(create)
    try fun(ctx) finally ctx.foreach(destroy)
  }

  /** Progress and abort of a `ProcessorImpl` as the callback of a native run; a run that the callback aborted
    * comes back as an `IllegalStateException("aborted")`, which is turned into `Processor.Aborted`. */
  def run[A](p: ProcessorImpl[_, _], setProgress: Double => Unit, check: () => Unit)(fun: Progress => A): A = {
    val obs = new Progress {
      def apply(frac: Double): Boolean = {
        setProgress(frac)
        try { check(); true } catch { case Processor.Aborted() => false }
      }
    }
    try fun(obs) catch {
      case e: IllegalStateException if e.getMessage == "aborted" => throw Processor.Aborted()
    }
  }
}
